#!/bin/bash
# HBM traffic of the hot kernels from PMC counters (separate passes, as MI355X_MICROARCH.md prescribes).
# usage (GPU box): bash tools/pmc.sh   -> gpurun_out/pmc/{fetch,write}/..., gpurun_out/pmc/pmc_traffic.json
out=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $out/fetch $out/write
cd /tmp && export TMPDIR=/tmp
for pass in fetch write; do
  ctr=FETCH_SIZE; [ $pass = write ] && ctr=WRITE_SIZE
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$pass -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_workload.py > $out/$pass/stdout.log 2> $out/$pass/err.log
  echo "$pass exit $?"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summarise.py $out
