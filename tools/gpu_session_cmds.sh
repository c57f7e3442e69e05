# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session X: kw_embed with coalesced row stores)
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_wide.py tests/test_gpu_r6.py -q -m gpu -x 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
for c in ycbv small; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$c -- python $GRAFT_REPO_ROOT/tools/bench_wide.py --mlp f16x2 --cases $c --steps 2 --no-grad > $O/ks_$c.log 2>&1
grep "^$c" $O/ks_$c.log | cut -c1-330
f=$(ls $O/ks_$c/*/*_kernel_stats.csv | head -1); cut -c1-150 $f | grep "embed\|Name"
done
