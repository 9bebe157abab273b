# skinny step kernel with A loaded straight from global memory in fragment order (-DSKF_DIRECT_A build) against the LDS-staged form
for lib in libxgate_hip_diag.so libxgate_hip_directa.so; do
  for prec in fp32 bf16; do
    XG_LIBRARY=/root/repo/controllable_xgating_amd/lib/$lib python bench.py --precision $prec --no-pmc --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib $prec', 'iteration', d['ms_per_step'], 'ms; step alone', r['avg_launch_us'], 'us, in situ', r['in_situ_us_per_step'])"
  done
done
