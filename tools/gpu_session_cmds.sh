R=$PWD
T=$R/neural_sim_nerf_amd/csrc/ab/libnsr_timing.so
rm -f $O/phase_timers.txt
for mlp in f16x2 bf16x3 fp32; do
  echo "== $mlp" >> $O/phase_timers.txt; NSR_MLP=$mlp NSR_LIB_PATH=$T V=32 timeout 120 python $R/tools/phase_timers.py 2>/dev/null >> $O/phase_timers.txt
done
cat $O/phase_timers.txt
timeout 600 python bench.py --cpu-sample-side 64 --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], json.dumps(d['roofline'])[:700])"
