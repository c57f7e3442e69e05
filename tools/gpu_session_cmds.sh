# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r05 session U: whole GPU suite on the final tree,
# smoke, the driver-shaped bench line, layered timing + kernel stats + PMC)
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -q -m gpu --durations=8 > $O/gpu_suite.log 2>&1; tail -12 $O/gpu_suite.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_final_kernel.json 2> $O/bench.err; python - <<PY
import json
d = json.loads(open("$O/bench_final_kernel.json").read().strip().splitlines()[-1])
print("value", d["value"], "kernel_ms", d["roofline"]["kernel_ms"])
print("layered", json.dumps(d["extra_workloads"].get("layered"))[:1500])
PY
timeout 600 python tools/bench_wide.py --cases ycbv,w512,d10w384,small --steps 2 2>&1 | grep -v "^{" | cut -c1-800 > $O/bench_wide.txt; cat $O/bench_wide.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_wide -- python $GRAFT_REPO_ROOT/tools/bench_wide.py --cases w512 --steps 1 > $O/stats_wide.log 2>&1
cp $(find $O/stats_wide -name "*kernel_stats.csv" | head -1) $O/kernel_stats_layered_w512.csv; head -8 $O/kernel_stats_layered_w512.csv | cut -c1-150
cd $GRAFT_REPO_ROOT
bash tools/pmc_wide.sh $O/pmc w512 200 > $O/pmc.log 2>&1
python - <<PY
import json
d = json.load(open("$O/pmc/pmc_summary.json"))
for k, v in d.items():
    if "gemm" in k: print(k, json.dumps(v["derived"]))
PY
