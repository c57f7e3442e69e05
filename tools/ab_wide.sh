#!/bin/bash
# Timing ablations of the layered renderer's bf16x3 GEMM (csrc/nsr_wide_b3.inc, -DNSRW_EXP_<X>: results are WRONG, only the time
# and the clock mean something): tools/ab_wide.sh <outfile> [case] -- one forward view per build, interleaved twice.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$1; CASE=${2:-w512}
: > $OUT
for rnd in 1 2; do
  for x in shipped ${WM2:-WM2} NOA NODMA NOSPLIT NOSTORE NOBAR; do
    lib=$R/neural_sim_nerf_amd/csrc/ab/libnsr_wide_$x.so
    [ $x = shipped ] && lib=$R/neural_sim_nerf_amd/csrc/libnsr.so
    unset NSRW_B3_WM; [ $x = WM2 ] && lib=$R/neural_sim_nerf_amd/csrc/libnsr.so && export NSRW_B3_WM=2
    [ -f $lib ] || continue
    NSR_LIB_PATH=$lib timeout 300 python $R/tools/bench_wide.py --mlp ${MLP:-bf16x3} --cases $CASE --steps 2 --no-grad 2>/dev/null | grep "^$CASE" | \
      python -c "import sys,json; l=sys.stdin.readline(); d=json.loads(l[l.index('{'):])['forward']; print('%-8s ms %8.2f  TF %6.1f  sclk %6.1f MHz  %6.1f W' % ('$x', d['ms_per_view'], d['algorithmic_TFLOPs'], d['power_and_clock']['sclk_MHz_mean'], d['power_and_clock']['socket_power_W_mean']))" | tee -a $OUT
  done
done
