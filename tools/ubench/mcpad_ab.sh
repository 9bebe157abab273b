#!/bin/bash
# m-contiguous LDS image padding (GEMM_MC_PAD 4 vs 8): half-waves of a ds_read_b32 fragment read on disjoint bank halves?
cd $GRAFT_REPO_ROOT
python - <<'PY'
import __graft_entry__ as g
print(g.build_variant("mcp8", g.FLAGS + ["-DGEMM_MC_PAD=8"]))
PY
for lib in libxgate_hip.so libxgate_hip_mcp8.so; do
  echo "== $lib"
  XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/$lib python tools/ubench/gemm_bench.py one 0 2>/dev/null | tail -1 | python -c "
import sys,json
l=sys.stdin.read(); d=json.loads(l[l.index('{'):])
for k,v in d.items(): print('  %-40s %8.1f us %7.1f TF  err %.1e'%(k,v[0],v[1],v[2]))"
done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in libxgate_hip.so libxgate_hip_mcp8.so; do
  rm -rf gpurun_out/pmc_mc
  XG_GEMM_SHAPES="wgrad TN" XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/$lib rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_mc -- python tools/ubench/gemm_bench.py one 0 > /dev/null 2>&1
  echo "== pmc $lib"
  python - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_mc/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"][:60]
        if "gemm" not in k: continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items():
    print("  ",k, {c: round(x/1e6,2) for c,x in v.items()})
PY
done
rm -rf gpurun_out/pmc_mc
