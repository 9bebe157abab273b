#!/bin/bash
# The failure table of docs/pkfma_hazard.md, end to end.
#   tools/r6/pkfma_table.sh build     (CPU, ~10 min: the nine variants of csrc/diag_pkfma_bisect.inc + the vectorizer-on product)
#   tools/r6/pkfma_table.sh run       (GPU: writes gpurun_out/pkfma_table.txt; copy into profiles/r06_pkfma_table.txt)
set -u
cd "$(dirname "$0")/../.."
L=controllable_xgating_amd/lib
case "${1:-run}" in
build)
    for n in 0 1 2 3 4 5 6 7 8; do
        extra=""; [ $n = 0 ] && extra=", '-fslp-vectorize'"
        python -c "import __graft_entry__ as g; g.build_variant('pkasm$n', ['-DSKF_PK_ASM=$n'$extra])" > /tmp/pkasm$n.log 2>&1 &
        [ $((n % 3)) = 2 ] && wait
    done
    python -c "import __graft_entry__ as g; g.build_variant('slponly', ['-fslp-vectorize'])" > /tmp/slponly.log 2>&1
    wait; ls -la $L/libxgate_hip_pkasm?.so $L/libxgate_hip_slponly.so ;;
run)
    mkdir -p gpurun_out; out=gpurun_out/pkfma_table.txt; : > $out
    reps=${2:-20}
    { echo "== product library"; python tools/r6/pkfma_runs.py $reps
      echo "== product sources, vectorizer on (-fslp-vectorize), context loop pinned"; XG_LIBRARY=$L/libxgate_hip_slponly.so python tools/r6/pkfma_runs.py $reps
      for n in 0 1 2 3 4 5 6 7 8; do
          echo "== variant $n ($(grep -m1 "^//   $n  \|    $n  " controllable_xgating_amd/csrc/diag_pkfma_bisect.inc | sed 's/^\/\/ *//'))"
          XG_LIBRARY=$L/libxgate_hip_pkasm$n.so python tools/r6/pkfma_runs.py $reps
      done
      echo "== where (variant 1, one step, 150 repetitions)"; XG_LIBRARY=$L/libxgate_hip_pkasm1.so python tools/r6/pkfma_where.py bf16x3 128 150 | grep -v identical
      echo "== where (variant 6, one step, 6 repetitions)"; XG_LIBRARY=$L/libxgate_hip_pkasm6.so python tools/r6/pkfma_where.py bf16x3 128 6 | grep -v identical
      echo "== stand-alone reproducer (tools/ubench/pkfma_repro.hip)"
      hipcc --offload-arch=gfx950 -O3 -o /tmp/pkfma_repro tools/ubench/pkfma_repro.hip && /tmp/pkfma_repro 30 300
    } 2>&1 | grep -v "amdgpu.ids" | cut -c1-400 >> $out
    tail -5 $out ;;
esac
