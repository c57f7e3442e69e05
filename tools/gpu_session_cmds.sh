mkdir -p $O/extra
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bf16x3 or selftest or native" 2>&1 | tail -15
NSR_MLP=bf16x3 timeout 100 python tools/one_view.py 16 0 phases 4 2>&1 | grep -v amdgpu.ids | tail -4
