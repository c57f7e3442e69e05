timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vjp or options or coarse_only or bf16x3" 2>&1 | grep -E "out of tolerance|passed|failed|Error|error|assert" | head -20
NSR_MLP=bf16x3 timeout 100 python tools/bench_vjp.py 400 3 2>&1 | grep -v amdgpu.ids | tail -2
