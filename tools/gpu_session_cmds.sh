set -x
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; echo rc $?; tail -c 3000 $O/bench.json; tail -5 $O/bench.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 > $O/bench_dist1.json 2> $O/bench_dist1.err; echo rc $?; tail -c 1500 $O/bench_dist1.json; tail -3 $O/bench_dist1.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 1 --workload sweep100 --views 6 > $O/bench_sweep.json 2> $O/bench_sweep.err; echo rc $?; tail -c 1500 $O/bench_sweep.json; tail -3 $O/bench_sweep.err
timeout 300 python bench.py --steps 1 --workload models21 > $O/bench_m21.json 2> $O/bench_m21.err; echo rc $?; tail -c 1500 $O/bench_m21.json; tail -3 $O/bench_m21.err
python bench.py --gpus 2 2>&1 | tail -1
