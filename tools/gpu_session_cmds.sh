# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session V: the round's profile collection)
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh 2>&1 | tail -5
