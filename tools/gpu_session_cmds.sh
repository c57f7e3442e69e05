# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r05 session F: where the parked cycles of the GEMM
# go -- ablations on top of the one-piece-per-step schedule; timing only, results of the ablated builds are wrong)
cd $GRAFT_REPO_ROOT
A=neural_sim_nerf_amd/csrc/ab
for v in t4 t4_nobar t4_nofrag t4_novm t4_nodma t4; do echo "== lib $v"; NSR_LIB_PATH=$A/libnsr_$v.so NSR_MLP=f16x2 timeout 200 python tools/phase_timers.py 2>&1 | grep -E "kernel ms|GEMMs|between"; done 2>&1 | tee $O/timers.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 -L 2>/dev/null | grep -iE "^\s*(SQ_|GRBM_)[A-Z_0-9]*" -o | sort -u | tr '\n' ' ') > $O/counters.txt; wc -c $O/counters.txt
