#!/bin/bash
# usage (GPU box): bash tools/prof_script.sh <tag> <script.py> [args...]  -> gpurun_out/prof_<tag>/{summary.txt,stdout.txt}
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
script=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $script "$@" > $out/stdout.txt 2> $out/err.log
echo "rocprof exit $?"
python3 - "$out" <<'PY'
import csv, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(f"{out}/p_kernel_stats.csv")))
with open(f"{out}/summary.txt", "w") as f:
    for r in rows[:30]:
        line = f"{r['Name'][:120]:120s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f}"
        print(line); f.write(line + "\n")
PY
