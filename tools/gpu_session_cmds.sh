timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; echo rc $?; python -c "
import json
d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['kernel'], d['roofline_vjp']['frac'], d['roofline_vjp']['kernel'], d['cpu_baseline']['value'], d['parity']['psnr_vs_oracle_db'], d['extra_workloads']['config1']['gpu']['value'])"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
