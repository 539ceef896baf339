#!/bin/bash
# HBM traffic + kernel times of the fused edge pass (separate passes: --pmc never together with other traces than kernel-trace).
# usage (GPU box): bash tools/pmc_conv.sh -> gpurun_out/pmc_conv/
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv
mkdir -p $out/fetch $out/write $out/trace
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_conv_workload.py > $out/trace/stdout.log 2> $out/trace/err.log
echo "trace exit $?"
for pass in fetch write; do
  ctr=FETCH_SIZE; [ $pass = write ] && ctr=WRITE_SIZE
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$pass -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_conv_workload.py > $out/$pass/stdout.log 2> $out/$pass/err.log
  echo "$pass exit $?"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summarise.py $out > $out/summary.txt
cat $out/summary.txt
python3 - "$out" <<'PY'
import csv, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(f"{out}/trace/p_kernel_stats.csv")))
with open(f"{out}/kernel_stats.txt", "w") as f:
    for r in rows[:12]:
        line = f"{r['Name'][:120]:120s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:10.1f}"
        print(line); f.write(line + "\n")
PY
