#!/bin/bash
# LDS bank-conflict counters of the GEMM kernels (PMC pass of its own, kernel-trace only).
# usage (GPU box): bash tools/pmc_gemm.sh -> gpurun_out/pmc_gemm/
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_gemm_workload.py > $out/stdout.log 2> $out/err.log
echo "pmc exit $?"
python3 - "$out" <<'PY'
import csv, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"{out}/p_counter_collection.csv")):
    acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{out}/summary.txt", "w") as f:
    for k, c in acc.items():
        if "gemm" not in k:
            continue
        conf = sum(c.get("SQ_LDS_BANK_CONFLICT", [0])) / max(len(c.get("SQ_LDS_BANK_CONFLICT", [1])), 1)
        act = sum(c.get("SQ_LDS_IDX_ACTIVE", [0])) / max(len(c.get("SQ_LDS_IDX_ACTIVE", [1])), 1)
        line = f"{k:90s} launches={len(c.get('SQ_LDS_IDX_ACTIVE', []))} LDS_BANK_CONFLICT={conf:.3e} LDS_IDX_ACTIVE={act:.3e} conflict_share={conf / max(act, 1):.4f}"
        print(line); f.write(line + "\n")
PY
