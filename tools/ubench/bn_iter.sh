#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | head -5
for i in 1 2 3; do python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xe', d['ms_per_step'], d['roofline'].get('in_situ_us_per_step'), d.get('final_loss'))"; done
python bench.py --no-secondary --no-cpu-baseline --no-pmc --workload scst 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('scst', d['ms_per_step'])"
python bench.py --no-secondary --no-cpu-baseline --no-pmc --workload xe5 --precision bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xe5', d['ms_per_step'])"
XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/libxgate_hip_diag.so XG_TD_ALL=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "weight_gradient_layout" 2>&1 | grep -E "passed|failed|rror" | head -3
