#!/bin/bash
# per-kernel times of tools/time_partition.py under rocprofv3 (kernel trace only). usage (GPU box): bash tools/prof_partition.sh <tag> [lib]
tag=$1; lib=$2
out=$GRAFT_REPO_ROOT/gpurun_out/prof_part_$tag
mkdir -p $out
[ -n "$lib" ] && export GNNOPS_LIB_PATH=$lib
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/time_partition.py > $out/stdout.txt 2> $out/err.log
echo "rocprof exit $?"
python3 - "$out" <<'PY'
import csv, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(f"{out}/p_kernel_stats.csv")))
with open(f"{out}/summary.txt", "w") as f:
    for r in rows[:24]:
        line = f"{r['Name'][:110]:110s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f} max_us={float(r['MaxNs'])/1e3:9.1f}"
        print(line); f.write(line + "\n")
PY
