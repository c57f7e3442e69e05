#!/bin/bash
# Timing ablations of the f16x2 input-gradient kernel k_render_vjp_h2 (the four ablations of
# profiles/r05/extra/sched_ab.txt, which were run on the forward kernel only).  ab/libnsr_h2_<X>.so = libnsr.so with the f16x2
# unit built -DNSR_EXP_<X> (results are WRONG, only time / clock / power mean something).  tools/ab_vjp_h2.sh <outfile>
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$1
: > $OUT
for rnd in 1 2; do
  for x in shipped NOBARRIER H2_NOFRAG NOVMWAIT NODMA; do
    lib=$R/neural_sim_nerf_amd/csrc/ab/libnsr_h2_$x.so
    [ $x = shipped ] && lib=$R/neural_sim_nerf_amd/csrc/libnsr.so
    [ -f $lib ] || continue
    NSR_LIB_PATH=$lib python - <<PY | tee -a $OUT
import sys, os, json, numpy as np, torch
sys.path.insert(0, "$R")
from neural_sim_nerf_amd import synthetic as S
from neural_sim_nerf_amd.engine import NsrModel
from bench import PowerSampler
sd_c = S.synth_weights(0); sd_f = S.synth_weights(1000, fine_of=sd_c)
m = NsrModel(sd_c, sd_f)
o, d = m.get_rays(400, 400, S.YCBV_K, S.sweep_poses(1, 0)[0])
o, d = o.reshape(-1, 3), d.reshape(-1, 3)
cot = torch.randn(160000, 3, device=m.device)
m.render_rays_vjp(o, d, S.YCBV_NEAR, S.YCBV_FAR, cot)
ps = PowerSampler(m.device.index); ps.start()
ms = []
for _ in range(6):
    m.render_rays_vjp(o, d, S.YCBV_NEAR, S.YCBV_FAR, cot); ms.append(m.last_kernel_ms())
p = ps.stop()
fw = []
for _ in range(6):
    m.render_rays(o, d, S.YCBV_NEAR, S.YCBV_FAR); fw.append(m.last_kernel_ms())
print("%-10s vjp ms %7.2f  sclk %6.1f MHz  %6.1f W   (forward kernel %6.2f ms)" % ("$x", float(np.median(ms)), p["sclk_MHz_mean"], p["socket_power_W_mean"], float(np.median(fw))))
PY
  done
done
