# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session AB: paired A loads, A/B)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_wide.py tests/test_gpu_r6.py -q -m gpu -x 2>&1 | tail -5
for rnd in 1 2; do
for x in shipped NOPAIR; do
lib=$GRAFT_REPO_ROOT/neural_sim_nerf_amd/csrc/ab/libnsr_wide_$x.so; [ $x = shipped ] && lib=$GRAFT_REPO_ROOT/neural_sim_nerf_amd/csrc/libnsr.so
for mlp in f16x2 bf16x3 fp32; do
for c in ycbv w512 small; do
[ $mlp = fp32 ] && [ $rnd = 2 ] && continue
echo "$x $mlp: $(NSR_LIB_PATH=$lib timeout 300 python tools/bench_wide.py --mlp $mlp --cases $c --steps 2 --no-grad 2>/dev/null | grep "^$c" | python -c "import sys,json; l=sys.stdin.readline(); d=json.loads(l[l.index('{'):]); f=d['forward']; print('%-36s ms %8.2f  TF %6.1f  sclk %6.1f MHz  %6.1f W' % (d['network'][:34], f['ms_per_view'], f['algorithmic_TFLOPs'], f['power_and_clock']['sclk_MHz_mean'], f['power_and_clock']['socket_power_W_mean']))")"
done
done
done
done 2>&1 | tee $O/pairload_ab.txt
