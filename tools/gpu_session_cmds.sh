timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 60 python tools/one_view.py 16 0 phases 4 2>&1 | grep -v amdgpu.ids
timeout 60 python tools/one_view.py 16 0 queue 4 2>&1 | grep -v amdgpu.ids
timeout 60 python tools/bench_vjp.py 400 3 2>&1 | grep -v amdgpu.ids
