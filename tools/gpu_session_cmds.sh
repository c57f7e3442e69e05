# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r05 session A: the round's new tests, an A/B of the
# kernel change against the r04 library, the box's power / clock files, one bench line without the CPU leg)
cd $GRAFT_REPO_ROOT
echo "== hwmon"; ls /sys/class/drm/ 2>&1 | head; for f in /sys/class/drm/card*/device/hwmon/hwmon*/{power1_average,power1_input,freq1_input,power1_cap}; do echo $f $(cat $f 2>&1); done
(timeout 20 amd-smi metric -p -c --json 2>&1 | head -40) > $O/amd_smi.txt; head -5 $O/amd_smi.txt
echo "== new tests"
timeout 1500 python -m pytest tests/test_gpu_r5.py tests/test_gpu_r4.py -q -m gpu -x --durations=15 > $O/new_tests.log 2>&1; tail -25 $O/new_tests.log
echo "== A/B r04 library vs this tree"
timeout 300 python tools/ab_h2.py --n 8 neural_sim_nerf_amd/csrc/ab/libnsr_r04.so neural_sim_nerf_amd/csrc/libnsr.so neural_sim_nerf_amd/csrc/ab/libnsr_r04.so neural_sim_nerf_amd/csrc/libnsr.so 2>&1 | tee $O/ab.txt
echo "== bench"
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json; tail -5 $O/bench.err
