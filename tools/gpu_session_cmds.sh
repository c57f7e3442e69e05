mkdir -p $O/extra
timeout 300 python bench.py --steps 3 --warmup 1 > $O/extra/bench.json 2> $O/extra/bench.err; tail -c 600 $O/extra/bench.json
