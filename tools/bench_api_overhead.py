"""Drop-in API overhead (bench.py: extra_workloads.api_overhead) on its own: python tools/bench_api_overhead.py
NSR_TRUST_VERSIONS=1 in front of it: the same without the per-call content fingerprint of the weights."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from neural_sim_nerf_amd import synthetic as S
sd_c = S.synth_weights(0)
r = bench.api_overhead_workload(sd_c, S.synth_weights(1000, fine_of=sd_c), 0)
r["NSR_TRUST_VERSIONS"] = os.environ.get("NSR_TRUST_VERSIONS", "0")
print(json.dumps(r))
