#!/bin/bash
# rocprofv3 PMC passes over six forms of tools/micro/store_sweep.bin (grid-strided vs contiguous-chunk copy / fill / 5:1 mix):
# memory-side request counts and stall counters per kernel. One counter group per pass (TCC has 4 slots).
# usage (on the GPU box, from the repo root): bash tools/pmc_store_sweep.sh gpurun_out/pmc_store
set -e
OUT=${1:-gpurun_out/pmc_store}
mkdir -p "$OUT"
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum" \
           "TCC_BUSY_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d "$ROOT/$OUT/p$i" -o p$i --output-format csv -- "$ROOT/tools/micro/store_sweep.bin" 4 1 pmc > "$ROOT/$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$ROOT/$OUT/p$i.log"; }
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tab = collections.OrderedDict()
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sweep_kernel" not in k: continue
        tab.setdefault(k, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, cs in tab.items():
        fh.write(k + "\n")
        for c, v in cs.items():
            fh.write(f"    {c:44s} {sum(v)/len(v):.6g}   (launches {len(v)})\n")
print(open(out + "/summary.txt").read())
PY
