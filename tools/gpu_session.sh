#!/bin/bash
# ad-hoc GPU session: args = label, then commands are read from tools/gpu_session_cmds.sh
R=${GRAFT_REPO_ROOT:-$PWD}
export O=$R/gpurun_out/$1
mkdir -p $O
cd $R
bash tools/gpu_session_cmds.sh 2>&1 | tee $O/session.log | tail -${2:-80}
