NSR_PROBE_VERBOSE=1 MODES=11,11,17,12,25,11,17,12,25 timeout 200 python tools/probe_bf16x3.py 2>&1 | grep -v amdgpu.ids | tail -18
