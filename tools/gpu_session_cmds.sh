timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c_host" 2>&1 | tail -12
