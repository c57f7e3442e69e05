# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session Y: whole GPU suite, smoke, bench on the final tree; layered re-collection)
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu -x --durations=8 2>&1 | tail -16
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "dtype")}, d["roofline"]["frac"], d["roofline"]["issued_frac"], d["parity_summary"])
print({k: (v.get("ms_per_view"), v.get("achieved")) for k, v in d["extra_workloads"]["layered"].items() if isinstance(v, dict)})
print(d["extra_workloads"]["trained"]["passes"], d["extra_workloads"]["api_overhead"]["layered_patch_512"])
PY
R=$GRAFT_REPO_ROOT; P=$R/gpurun_out/prof; mkdir -p $P
for mlp in f16x2 bf16x3 fp32; do
  timeout 900 python tools/bench_wide.py --mlp $mlp --cases ycbv,w512,d10w384,small,w1024 --steps 2 2>/dev/null | grep -v "^{" > $P/layered_bench_$mlp.txt
done
cd /tmp && export TMPDIR=/tmp
for c in ycbv w512; do
  rm -rf $P/stats_layered_$c
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_layered_$c -- python $R/tools/bench_wide.py --mlp f16x2 --cases $c --steps 2 > $P/stats_layered_$c.log 2>&1
done
cut -c1-260 $P/layered_bench_f16x2.txt
