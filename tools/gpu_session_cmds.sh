MODES=11,16,12,11,16,12,11,16 timeout 120 python tools/probe_bf16x3.py 2>&1 | grep -v amdgpu.ids | tail -9
