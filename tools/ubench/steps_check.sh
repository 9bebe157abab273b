B="python bench.py --no-cpu-baseline --no-pmc --no-secondary"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"])'
$B --precision bf16x3 --steps 10 --warmup 5 2>/dev/null | python -c "$P" "x3 10/5"
$B --precision bf16x3 --steps 20 --warmup 5 2>/dev/null | python -c "$P" "x3 20/5"
$B --steps 10 --warmup 5 2>/dev/null | python -c "$P" "fp32 10/5"
$B --steps 20 --warmup 5 2>/dev/null | python -c "$P" "fp32 20/5"
$B --workload xe5 --precision bf16 --steps 10 --warmup 5 2>/dev/null | python -c "$P" "xe5 10/5"
$B --workload scst --steps 10 --warmup 5 2>/dev/null | python -c "$P" "scst 10/5"
