#!/bin/bash
# usage (GPU box): bash tools/prof_gemm_sweep.sh <tag> [L ...] -> per-length kernel times (pad copies vs product)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_gemm_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_gemm_sweep.py "$@" > $out/run.log 2> $out/err.log
echo "rocprof exit $?"
cat $out/run.log
python3 - "$out" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(f"{out}/**/p_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# group by (kernel, grid) and report mean duration
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if not ("gemm" in n or "pad_rows" in n): continue
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    key = (n.split("(")[0][:60], r["Grid_Size_X"], r["Grid_Size_Y"])
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    v = v[len(v) // 4:]
    print(f"{k[0]:60s} grid=({k[1]},{k[2]}) n={len(v):3d} avg_us={sum(v)/len(v):9.1f}")
PY
