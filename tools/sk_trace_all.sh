#!/bin/bash
# in-kernel phase timelines of the three launches of the decoder step (B = 128).  Build the trace variant first (here or on the box):
#   python -c "import __graft_entry__ as g; g.build_variant('sktrace', ['-DSK_TRACE'])"
cd ${GRAFT_REPO_ROOT:-.}
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_sktrace.so
[ -f $XG_LIBRARY ] || python -c "import __graft_entry__ as g; g.build_variant('sktrace', ['-DSK_TRACE'])"
for prec in ${SK_PRECS:-fp32}; do SK_PREC=$prec timeout 400 python tools/sk_trace_run.py 2>&1 | grep -v Warning; done
