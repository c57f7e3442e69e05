MODES=11,11,16,17,11,16,17,11,16 timeout 200 python tools/probe_bf16x3.py 2>&1 | grep -v amdgpu.ids | tail -9
