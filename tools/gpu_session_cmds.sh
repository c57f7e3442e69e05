# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session AO: networks the fused kernels serve by re-expression -- is the layered renderer cheaper?)
cd $GRAFT_REPO_ROOT
timeout 900 python tools/bench_wide.py --mlp f16x2 --cases d8w64,d8w128,d8w192,d4w256 --steps 2 2>/dev/null | grep -v "^{" > $O/narrow.txt
python - <<PY
import json
for l in open("$O/narrow.txt"):
    d = json.loads(l[l.index("{"):]); f = d["forward"]; g = d["forward+input-gradient"]
    print("%-34s fwd %8.2f ms | fwd+grad %8.2f ms   (fused kernels, any re-expressed network: 108 / 200 ms)" % (d["network"][:32], f["ms_per_view"], g["ms_per_view"]))
PY
