#!/bin/bash
# usage (GPU box): bash tools/prof_big.sh <tag> <name substring ...> -> kernel stats of tools/big_shapes.py for those lines
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_big_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/big_shapes.py "$@" > $out/run.log 2> $out/err.log
echo "rocprof exit $?"
cat $out/run.log
python3 - "$out" <<'PY'
import csv, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(f"{out}/p_kernel_stats.csv")))
for r in rows[:12]:
    print(f"{r['Name'][:120]:120s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f}")
PY
