#!/bin/bash
# usage (GPU box): bash tools/prof_conv.sh <layer> <batch size>  -> gpurun_out/prof_conv_<layer>_<bs>/
out=$GRAFT_REPO_ROOT/gpurun_out/prof_conv_$1_$2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_conv.py $1 $2 > $out/run.log 2> $out/err.log
echo "rocprof exit $?"; cat $out/run.log
python3 - "$out" <<'PY'
import csv, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(f"{out}/p_kernel_stats.csv")))
with open(f"{out}/summary.txt", "w") as f:
    for r in rows[:40]:
        line = f"{r['Name'][:110]:110s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.2f}"
        print(line); f.write(line + "\n")
PY
