#!/bin/bash
# usage (GPU box): bash tools/prof_op.sh <tag> <ops> [point]  -> kernel stats of benchmark_ops.py for those ops
tag=$1; ops=$2; point=${3:-ref_max}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/gnn-ops-benchmark_amd/op_bm_scripts/benchmark_ops.py --ops $ops --point $point --runs 10 > $out/run.log 2> $out/err.log
echo "rocprof exit $?"
python3 - "$out" <<'PY'
import csv, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(f"{out}/p_kernel_stats.csv")))
for r in rows[:14]:
    print(f"{r['Name'][:110]:110s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f}")
PY
grep "L=" $out/run.log
