# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session K: compact raw layout + fast encodings)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_wide.py -q -m gpu 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
for c in ycbv small; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$c -- python $GRAFT_REPO_ROOT/tools/bench_wide.py --mlp bf16x3 --cases $c --steps 2 > $O/ks_$c.log 2>&1
grep "^$c" $O/ks_$c.log | cut -c1-400
f=$(ls $O/ks_$c/*/*_kernel_stats.csv | head -1); cut -c1-130 $f | grep -v "kw_gemm" | head -9
done
