#!/bin/bash
# The 17 reference scripts' OWN sweeps at full length (100-800 points each, the reference's timer protocol), caches off (cold):
# writes the reference's CSVs under $1 (default gpurun_out/ref_sweeps_full). One script at a time, 15 min each at most.
# --num 100: the reference's PUBLISHED CSVs have 100 lengths per sweep (mem_prof_data/native_index_select.csv: 800 rows) where the
# current text of six scripts says num=10 (benchmark_native_index_select.py:38).
out=${1:-gpurun_out/ref_sweeps_full}
mkdir -p "$out"
cd gnn-ops-benchmark_amd
for op in scatter_add scatter_min scatter_max scatter_mean native_index_select native_index_add_ native_gather sparse_transpose \
          fused_index_select_reduce fused_index_add_reduce native_addmm native_matmul sparse_spmm sparse_spspmm scatter_multiply \
          native_sort sparse_coalesce; do
  start=$(date +%s)
  PYTHONPATH=. timeout -k 10 900 python op_bm_scripts/benchmark_ops.py --sweep ref --ops $op --num 100 --out ../$out > ../$out/$op.log 2>&1
  echo "$op rc=$? $(( $(date +%s) - start )) s  $(grep -c 'done with' ../$out/$op.log) points" | tee -a ../$out/summary.txt
done
