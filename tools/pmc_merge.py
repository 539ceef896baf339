"""Merge the kernels bench.py quotes from gpurun_out/pmc/pmc_traffic_raw.json into profiles/pmc_traffic.json."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = json.load(open(os.path.join(ROOT, "gpurun_out", "pmc", "pmc_traffic_raw.json")))
dst_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
dst = json.load(open(dst_path))
want = {"bucket_reduce_kernel<float, 0>": "bucket_reduce_kernel_f32_sum", "bucket_reduce_kernel<float, 2>": "bucket_reduce_kernel_f32_min",
        "seg_rows_kernel<float, 0>": "seg_rows_kernel_f32_sum", "seg_rows_kernel<float, 2>": "seg_rows_kernel_f32_min"}
for k, v in raw.items():
    for pat, name in want.items():
        if k.startswith(pat):
            dst[name] = v
            print(name, round(v["hbm_bytes_per_launch"] / 1e9, 3), "GB")
json.dump(dst, open(dst_path, "w"), indent=1)
