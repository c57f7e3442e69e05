timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "phases" 2>&1 | tail -3
timeout 120 python tools/one_view.py 16 0 phases 4 2>&1 | grep -v amdgpu.ids
timeout 120 python tools/one_view.py 16 0 queue 4 2>&1 | grep -v amdgpu.ids
