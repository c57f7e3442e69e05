# final collection of the round
bash tools/collect_profiles.sh > $O/collect.log 2>&1; tail -3 $O/collect.log
R=$PWD
mkdir -p gpurun_out/prof/extra
timeout 300 python bench.py --gpus 2 --backend gloo --share-gpu --workload sweep100 --views 12 --steps 1 --warmup 1 2>/dev/null | tail -1 > gpurun_out/prof/extra/bench_2ranks_shared_gpu_sweep12.json
timeout 300 python bench.py --workload sweep100 --views 24 --steps 1 --warmup 1 2>/dev/null | tail -1 > gpurun_out/prof/extra/bench_sweep24.json
timeout 300 python bench.py --workload models21 --steps 1 --warmup 1 2>/dev/null | tail -1 > gpurun_out/prof/extra/bench_models21.json
for f in gpurun_out/prof/extra/*.json; do echo $f; cut -c1-400 $f; done
