# what the last GPU session of round 3 ran (scratch file: tools/gpu_session.sh <label> executes it on the gpurun box)
timeout 1800 python -m pytest tests/ -q -m gpu 2>&1 | tail -4
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
