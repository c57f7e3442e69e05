# scratch command list for tools/gpu_session.sh (edited per GPU session)
R=$PWD
export NSR_MLP=f16x2
for lib in neural_sim_nerf_amd/csrc/libnsr.so neural_sim_nerf_amd/csrc/ab/libnsr_csplit.so neural_sim_nerf_amd/csrc/ab/libnsr_nobarrier.so neural_sim_nerf_amd/csrc/libnsr.so; do
  echo "== $lib"; NSR_LIB_PATH=$R/$lib timeout 120 python tools/one_view.py 32 0 queue 4 2>&1 | tail -3
done
echo "== phase timers f16x2"; NSR_LIB_PATH=$R/neural_sim_nerf_amd/csrc/ab/libnsr_timing.so V=32 timeout 120 python tools/phase_timers.py 2>&1 | tail -14
echo "== phase timers bf16x3"; NSR_MLP=bf16x3 NSR_LIB_PATH=$R/neural_sim_nerf_amd/csrc/ab/libnsr_timing.so V=32 timeout 120 python tools/phase_timers.py 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_h2_$i -- python $R/tools/one_view.py 32 > $O/pmc_h2_$i.log 2>&1
done
cd $R
python - <<'P'
import csv, glob, os
O=os.environ["O"]
tot={}; ns=None
for d in sorted(glob.glob(O+"/pmc_h2_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d+"/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_render_h2" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
    for f in glob.glob(d+"/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "k_render_h2" in r["Kernel_Name"] and ns is None:
                ns=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
print(tot, ns)
if "GRBM_GUI_ACTIVE" in tot and ns:
    print("clock GHz", tot["GRBM_GUI_ACTIVE"]/8/ns, "mfma busy", tot["SQ_VALU_MFMA_BUSY_CYCLES"]/(1024*tot["GRBM_GUI_ACTIVE"]/8))
if "SQ_INSTS_MFMA" in tot:
    print("valu per mfma", (tot["SQ_INSTS_VALU"]-tot["SQ_INSTS_MFMA"])/tot["SQ_INSTS_MFMA"], "lds per mfma", tot["SQ_INSTS_LDS"]/tot["SQ_INSTS_MFMA"], "salu per mfma", tot["SQ_INSTS_SALU"]/tot["SQ_INSTS_MFMA"])
if "SQ_WAIT_ANY" in tot:
    print("wait_any/active_any", tot["SQ_WAIT_ANY"], tot["SQ_WAIT_INST_ANY"], tot["SQ_ACTIVE_INST_ANY"], tot["SQ_ACTIVE_INST_VALU"])
P
unset NSR_MLP
timeout 1500 python -m pytest tests/ -q -m gpu -k "bf16x3_stagewise or f16x2 or census" -s 2>&1 | grep -v "^$" | grep -E "^census|raw0 max|passed|failed|FAILED|Error|assert" | cut -c1-900 | tail -30
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'P'
import json,os
d=json.loads(open(os.environ["O"]+"/bench.json").read().strip().splitlines()[-1])
print("api_overhead", json.dumps(d["extra_workloads"]["api_overhead"]))
P
