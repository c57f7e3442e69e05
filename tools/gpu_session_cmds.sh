# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session AC: 128-row tiles (two workgroups per CU) against 256-row tiles on the narrow networks)
cd $GRAFT_REPO_ROOT
for rnd in 1 2; do
for wm in 4 2; do
for mlp in f16x2 bf16x3; do
for c in ycbv small d10w384; do
echo "WM=$wm $mlp: $(NSRW_B3_WM=$wm timeout 300 python tools/bench_wide.py --mlp $mlp --cases $c --steps 2 2>/dev/null | grep "^$c" | python -c "import sys,json; l=sys.stdin.readline(); d=json.loads(l[l.index('{'):]); f=d['forward']; g=d['forward+input-gradient']; print('%-36s fwd ms %8.2f  TF %6.1f  sclk %6.1f MHz  %6.1f W | fwd+grad ms %8.2f' % (d['network'][:34], f['ms_per_view'], f['algorithmic_TFLOPs'], f['power_and_clock']['sclk_MHz_mean'], f['power_and_clock']['socket_power_W_mean'], g['ms_per_view']))")"
done
done
done
done 2>&1 | tee $O/wm_ab.txt
