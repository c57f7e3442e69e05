mkdir -p $O/extra
timeout 300 python bench.py --workload sweep100 --views 24 --steps 1 --warmup 0 > $O/extra/bench_sweep24.json 2> $O/extra/sweep.err; tail -c 300 $O/extra/bench_sweep24.json; echo
timeout 300 python bench.py --workload models21 --steps 1 --warmup 1 > $O/extra/bench_models21.json 2> $O/extra/models.err; tail -c 300 $O/extra/bench_models21.json; echo
NSR_DIST_TIMING=1 timeout 400 python bench.py --gpus 2 --backend gloo --share-gpu --workload sweep100 --views 12 --steps 1 --warmup 0 > $O/extra/bench_2ranks_shared_gpu_sweep12.json 2> $O/extra/ranks.err; tail -c 400 $O/extra/bench_2ranks_shared_gpu_sweep12.json; echo
NSR_MLP=f16x2 timeout 200 python tools/bench_path_grad.py > $O/extra/path_grad_f16x2.json 2> /dev/null; cat $O/extra/path_grad_f16x2.json | head -3
