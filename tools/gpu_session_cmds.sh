# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session AD: final-tree bench line + kernel stats; PMC of kw_gemm_f32)
R=$GRAFT_REPO_ROOT; P=$R/gpurun_out/prof; mkdir -p $P
cd $R && python -c "from neural_sim_nerf_amd import _lib; print(_lib.kernel_source_hash())" > $P/kernel_source_sha256.txt
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 3 --warmup 1 > $P/bench.json 2> $P/bench.err
rm -rf $P/stats_f16x2; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_f16x2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --mlp f16x2 > $P/stats_f16x2.log 2>&1
for c in ycbv w512; do
  rm -rf $P/stats_layered_$c
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_layered_$c -- python $R/tools/bench_wide.py --mlp f16x2 --cases $c --steps 2 > $P/stats_layered_$c.log 2>&1
done
cd $R
rm -rf $P/layered_pmc_fp32_w512; MLP=fp32 bash tools/pmc_wide.sh $P/layered_pmc_fp32_w512 w512 200 > /dev/null 2>&1
python - <<PY
import json
d = json.loads(open("$P/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "dtype")}, d["roofline"]["frac"], d["roofline"]["issued_frac"], d["parity_summary"])
print({k: (v.get("ms_per_view"), v.get("achieved"), v.get("frac")) for k, v in d["extra_workloads"]["layered"].items() if isinstance(v, dict)})
j = json.load(open("$P/layered_pmc_fp32_w512/pmc_summary.json"))
print({k: v["derived"] for k, v in j.items() if "gemm_f32<4, 1" in k})
PY
