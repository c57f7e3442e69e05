# round 4, session j: everything profiles/r04 holds (tools/collect_profiles.sh)
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh 2>&1 | tail -5
cat gpurun_out/prof/importance_counts.txt
tail -c 600 gpurun_out/prof/bench.json
