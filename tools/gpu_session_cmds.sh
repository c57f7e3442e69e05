# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session AQ: layered bench files on the final tree)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; P=$R/gpurun_out/prof; mkdir -p $P
for mlp in f16x2 bf16x3 fp32; do
  timeout 900 python tools/bench_wide.py --mlp $mlp --cases ycbv,w512,d10w384,small,w1024 --steps 2 2>/dev/null | grep -v "^{" > $P/layered_bench_$mlp.txt
  python - <<PY
import json
for l in open("$P/layered_bench_$mlp.txt"):
    d = json.loads(l[l.index("{"):]); f = d["forward"]; g = d["forward+input-gradient"]
    print("%-8s %-34s fwd %8.2f ms %6.1f TF | fwd+grad %8.2f ms %6.1f TF" % ("$mlp", d["network"][:32], f["ms_per_view"], f["algorithmic_TFLOPs"], g["ms_per_view"], g["algorithmic_TFLOPs"]))
PY
done
