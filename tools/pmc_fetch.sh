#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one 400x400 view with the default kernel for several NSR_CHUNK values.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/fetch; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    NSR_CHUNK=$c timeout 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/c${c}_$ctr -- python $R/tools/one_view.py 16 $WG > $O/c${c}_$ctr.log 2>&1
  done
done
python - <<'PY'
import csv, glob, os
O=os.environ.get("GRAFT_REPO_ROOT", os.getcwd())+"/gpurun_out/fetch"
for d in sorted(glob.glob(O+"/c*_*")):
    if not os.path.isdir(d): continue
    tot=0
    for f in glob.glob(d+"/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_render16" in r["Kernel_Name"]: tot+=float(r["Counter_Value"])
    print(os.path.basename(d), "%.3f GB (counter KiB sum * 1024%s)" % (tot*1024*(2 if "FETCH" in d else 1)/1e9, ", doubled" if "FETCH" in d else ""))
PY
