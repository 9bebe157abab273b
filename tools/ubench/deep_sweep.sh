cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5m
bash tools/ubench/ab_stepgroup.sh "libxgate_hip_diag.so XG_SK_DEEP=0" "libxgate_hip_diag.so XG_SK_DEEP=1" "libxgate_hip_diag.so XG_SK_DEEP=2" "libxgate_hip_diag.so XG_SK_DEEP=3" > gpurun_out/r5m/ab.txt 2>&1; cat gpurun_out/r5m/ab.txt
run() { XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['ms_per_step'], r['avg_launch_us'], r['in_situ_us_per_step'])"; }
for i in 1 2; do for d in 0 1 3 7; do XG_SK_DEEP=$d run deep$d; done; done
