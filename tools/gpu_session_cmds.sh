# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r05 session O: timing of the layered renderer)
cd $GRAFT_REPO_ROOT
timeout 600 python tools/bench_wide.py --cases ycbv,w512,d10w384,small > $O/bench_wide.txt 2>&1; tail -6 $O/bench_wide.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_wide -o wide -- python $GRAFT_REPO_ROOT/tools/bench_wide.py --cases w512 --steps 2 > $O/prof_wide.log 2>&1
find $O/prof_wide -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_wide_w512.csv
head -14 $O/kernel_stats_wide_w512.csv | cut -c1-160
