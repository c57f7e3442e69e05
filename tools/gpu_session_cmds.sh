timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "phases or vjp" 2>&1 | tail -6
timeout 60 python tools/bench_vjp.py 400 3 2>&1 | grep -v amdgpu.ids
