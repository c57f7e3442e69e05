# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session AP: 192-column tiles)
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_wide.py tests/test_gpu_r6.py -q -m gpu -x 2>&1 | tail -5
for mlp in f16x2 bf16x3 fp32; do
timeout 900 python tools/bench_wide.py --mlp $mlp --cases d10w384,d8w192 --steps 2 2>/dev/null | grep -v "^{" > $O/nj3_$mlp.txt
python - <<PY
import json
for l in open("$O/nj3_$mlp.txt"):
    d = json.loads(l[l.index("{"):]); f = d["forward"]; g = d["forward+input-gradient"]
    print("%-8s %-34s fwd %8.2f ms %6.1f TF | fwd+grad %8.2f ms %6.1f TF" % ("$mlp", d["network"][:32], f["ms_per_view"], f["algorithmic_TFLOPs"], g["ms_per_view"], g["algorithmic_TFLOPs"]))
PY
done
