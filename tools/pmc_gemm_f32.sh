#!/bin/bash
# Matrix-pipe utilisation and effective clock of the fp32 GEMM kernels (ours and the library's) at 8192^3.
# usage (GPU box): bash tools/pmc_gemm_f32.sh -> gpurun_out/pmc_gemm_f32/summary.txt
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_f32
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_gemm_f32_workload.py > $out/stdout.log 2> $out/err.log
echo "pmc exit $?"
python3 - "$out" <<'PY'
import csv, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f"{out}/p_counter_collection.csv")):
    acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for r in csv.DictReader(open(f"{out}/p_kernel_trace.csv")):
    dur[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(f"{out}/summary.txt", "w") as f:
    for k, c in acc.items():
        if "gemm" not in k.lower() and "Cijk" not in k:
            continue
        mean = lambda n: sum(c.get(n, [0])) / max(len(c.get(n, [1])), 1)
        us = sum(dur[k]) / max(len(dur[k]), 1)
        gui = mean("GRBM_GUI_ACTIVE")
        line = (f"{k:90s} launches={len(dur[k])} avg_us={us:.1f} eff_clock_GHz={gui / 8 / us / 1e3:.3f} "
                + " ".join(f"{n}={mean(n):.4e}" for n in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")))
        print(line); f.write(line + "\n")
PY
