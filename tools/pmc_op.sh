#!/bin/bash
# usage (GPU box): bash tools/pmc_op.sh <tag> <ops> "<counters>" [kernel-substring]
tag=$1; ops=$2; ctrs=$3; pat=${4:-kernel}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/gnn-ops-benchmark_amd/op_bm_scripts/benchmark_ops.py --ops $ops --point ref_max --runs 3 > $out/run.log 2> $out/err.log
echo "pmc exit $?"
python3 - "$out" "$pat" <<'PY'
import csv, sys, collections
out, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"{out}/p_counter_collection.csv")):
    if pat in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k, {n: f"{sum(v)/len(v):.3e} (n={len(v)})" for n, v in c.items()})
PY
