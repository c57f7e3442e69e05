"""One 400x400 view with the chosen forward kernel: python tools/one_view.py <variant 16|32> [max_workgroups] [queue|phases] [n_launches]."""
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from neural_sim_nerf_amd import synthetic as S
from neural_sim_nerf_amd.engine import NsrModel
v = int(sys.argv[1]); wg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sd_c = S.synth_weights(0); sd_f = S.synth_weights(1000, fine_of=sd_c)
sched = (sys.argv[3] or None) if len(sys.argv) > 3 else None
n = int(sys.argv[4]) if len(sys.argv) > 4 else 1
m = NsrModel(sd_c, sd_f, variant=v, max_workgroups=wg, schedule=sched)      # variant 0 and no schedule: the engine's default ($NSR_MLP)
for _ in range(n):
    m.render_views(S.sweep_poses(1, 0)[0], 400, 400, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR)
    print("variant", v, "mlp", m.mlp, "wg", wg, "schedule", m.schedule, "ms", m.last_kernel_ms())
