timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "rccl" 2>&1 | tail -25
