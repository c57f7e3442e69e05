# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session AF: 64 GiB default workspace -- tests, smoke, layered bench)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_wide.py tests/test_gpu_r6.py tests/test_gpu_r5.py -q -m gpu -x 2>&1 | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
R=$GRAFT_REPO_ROOT; P=$R/gpurun_out/prof; mkdir -p $P
for mlp in f16x2 bf16x3 fp32; do
  timeout 900 python tools/bench_wide.py --mlp $mlp --cases ycbv,w512,d10w384,small,w1024 --steps 2 2>/dev/null | grep -v "^{" > $P/layered_bench_$mlp.txt
  python - <<PY
import json
for l in open("$P/layered_bench_$mlp.txt"):
    d = json.loads(l[l.index("{"):]); f = d["forward"]; g = d["forward+input-gradient"]
    print("%-10s %-34s fwd %8.2f ms %6.1f TF chunks %2d | fwd+grad %8.2f ms %6.1f TF chunks %2d" % ("$mlp", d["network"][:32], f["ms_per_view"], f["algorithmic_TFLOPs"], f["chunks"], g["ms_per_view"], g["algorithmic_TFLOPs"], g["chunks"]))
PY
done
