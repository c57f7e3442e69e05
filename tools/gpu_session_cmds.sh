timeout 90 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "phases" 2>&1 | tail -6
timeout 60 python tools/one_view.py 16 0 phases 4 2>&1 | grep -v amdgpu.ids
