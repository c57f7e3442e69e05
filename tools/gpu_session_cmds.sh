# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session S: staggered workgroup starts, A/B)
cd $GRAFT_REPO_ROOT
for rnd in 1 2; do
for x in shipped STAGGER; do
lib=$GRAFT_REPO_ROOT/neural_sim_nerf_amd/csrc/ab/libnsr_wide_$x.so; [ $x = shipped ] && lib=$GRAFT_REPO_ROOT/neural_sim_nerf_amd/csrc/libnsr.so
for c in w512 ycbv w1024; do
echo "$x: $(NSR_LIB_PATH=$lib timeout 300 python tools/bench_wide.py --mlp f16x2 --cases $c --steps 2 --no-grad 2>/dev/null | grep "^$c" | cut -c1-90,150-420)"
done
done
done 2>&1 | tee $O/stagger.txt
