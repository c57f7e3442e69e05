timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "phases or vjp or full_size or graph" 2>&1 | tail -5
timeout 60 python tools/one_view.py 16 0 phases 3 2>&1 | grep -v amdgpu.ids
