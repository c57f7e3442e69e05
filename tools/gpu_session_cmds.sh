# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r05 session AE: the layered renderer at W = 1024)
cd $GRAFT_REPO_ROOT
timeout 200 python tools/bench_wide.py --cases w1024 --steps 1 --no-grad 2>&1 | grep -v "^{" | cut -c1-500
