# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r05 session T: swizzled LDS tiles, 3 vs 4 workgroups per CU)
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_wide.py -x -q > $O/wide.log 2>&1; tail -3 $O/wide.log
for w in 4 3; do
  echo "== NSRW_GEMM_WGS=$w"
  NSRW_GEMM_WGS=$w timeout 600 python tools/bench_wide.py --cases ycbv,w512,small --steps 2 2>&1 | grep -v "^{" | cut -c1-700
done > $O/bench_wide_ab.txt 2>&1
cat $O/bench_wide_ab.txt
bash tools/pmc_wide.sh $O/pmc w512 200 > $O/pmc.log 2>&1
python - <<PY
import json
d = json.load(open("$O/pmc/pmc_summary.json"))
for k, v in d.items():
    if "gemm" in k: print(k, json.dumps(v["derived"]))
PY
