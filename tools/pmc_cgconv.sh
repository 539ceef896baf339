#!/bin/bash
# Instruction mix and stall counters of the cgconv edge pass. usage (GPU box): bash tools/pmc_cgconv.sh -> gpurun_out/pmc_cgconv/summary.txt
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_cgconv
mkdir -p $out/a $out/b
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace --output-format csv -d $out/a -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_cgconv_workload.py > $out/a/stdout.log 2> $out/a/err.log
echo "pass a exit $?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $out/b -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_cgconv_workload.py > $out/b/stdout.log 2> $out/b/err.log
echo "pass b exit $?"
python3 - "$out" <<'PY'
import csv, sys, collections, glob
out = sys.argv[1]
with open(f"{out}/summary.txt", "w") as f:
    for sub in ("a", "b"):
        acc = collections.defaultdict(list)
        for fn in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(fn)):
                if "edge_reduce_kernel" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            line = f"{k:24s} {sum(v) / len(v):.4e}  (n={len(v)})"
            print(line); f.write(line + "\n")
PY
