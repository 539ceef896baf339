#!/bin/bash
# Matrix-pipe utilisation and effective clock of the GEMM kernels at 8192^3 bf16 (PMC pass of its own, kernel-trace only).
# usage (GPU box): bash tools/pmc_gemm_mfma.sh -> gpurun_out/pmc_gemm_mfma/summary.txt
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_mfma
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_gemm_workload.py > $out/stdout.log 2> $out/err.log
echo "pmc exit $?"
python3 - "$out" <<'PY'
import csv, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f"{out}/p_counter_collection.csv")):
    acc[r["Kernel_Name"][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for r in csv.DictReader(open(f"{out}/p_kernel_trace.csv")):
    dur[r["Kernel_Name"][:80]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(f"{out}/summary.txt", "w") as f:
    for k, c in acc.items():
        if "gemm" not in k.lower() and "Cijk" not in k:
            continue
        mean = lambda n: sum(c.get(n, [0])) / max(len(c.get(n, [1])), 1)
        us = sum(dur[k]) / max(len(dur[k]), 1)
        gui, busy, cu, mops = mean("GRBM_GUI_ACTIVE"), mean("SQ_VALU_MFMA_BUSY_CYCLES"), mean("SQ_BUSY_CU_CYCLES"), mean("SQ_INSTS_VALU_MFMA_MOPS_F32")
        line = (f"{k:80s} launches={len(dur[k])} avg_us={us:.1f} GRBM_GUI_ACTIVE={gui:.4e} (eff clock {gui / 8 / us / 1e3:.2f} GHz if summed over 8 XCDs) "
                f"MFMA_BUSY_CYCLES={busy:.4e} BUSY_CU_CYCLES={cu:.4e} MFMA_MOPS_F32={mops:.4e}")
        print(line); f.write(line + "\n")
PY
