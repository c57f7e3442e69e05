# scratch command list for tools/gpu_session.sh (edited per GPU session); default: the GPU suite, smoke and one bench line
timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -2
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python bench.py 2>/dev/null | tail -1 | cut -c1-400
