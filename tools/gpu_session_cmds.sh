# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session AE: workspace cap above 16 GiB)
cd $GRAFT_REPO_ROOT
for gb in 16 32 64; do
for c in ycbv w512; do
echo "ws $gb GB: $(NSR_WIDE_WORKSPACE_GB=$gb timeout 300 python tools/bench_wide.py --mlp f16x2 --cases $c --steps 2 2>/dev/null | grep "^$c" | python -c "import sys,json; l=sys.stdin.readline(); d=json.loads(l[l.index('{'):]); f=d['forward']; g=d['forward+input-gradient']; print('%-34s fwd ms %8.2f chunks %3d | fwd+grad ms %8.2f chunks %3d ws %.1f GB' % (d['network'][:32], f['ms_per_view'], f['chunks'], g['ms_per_view'], g['chunks'], g['workspace_GB']))")"
done
done 2>&1 | tee $O/ws_big.txt
