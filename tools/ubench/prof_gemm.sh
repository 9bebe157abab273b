#!/bin/bash
# kernel-trace of the GEMM ubench (guarded: never reads stdin)
cd /tmp; export TMPDIR=/tmp
GEMM_RANDOM=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- /root/repo/tools/ubench/gemm_ubench_BASE > /tmp/rp.log 2>&1 < /dev/null
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-170 "$f" | head -8
