# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r05 session J: the isolated f16x2 layer, libnsr_probe modes 28-30)
cd $GRAFT_REPO_ROOT
timeout 300 python tools/probe_h2.py 2>&1 | tee $O/probe_h2.txt
