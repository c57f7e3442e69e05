timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fewer_importance" 2>&1 | tail -25
