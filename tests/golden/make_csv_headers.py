"""Generate tests/golden/csv_headers.json: for each of the reference's 17 op scripts, the CSV path it writes and the header
row of that CSV — DATA the reference's scripts hold (their `df.columns = [...]` literal, also where the script has it
commented out, and the `to_csv(...)` path), plus, for benchmark_native_sort.py, the header the reference's own DataWriter
class writes (imported from /root/reference and run, as SURVEY.md §8c records it can be).

Run in the build container (the GPU box has no /root/reference):  python tests/golden/make_csv_headers.py
"""
import ast
import csv
import json
import os
import re
import sys
import tempfile

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csv_headers.json")


def columns_and_path(text):
    body = "\n".join(re.sub(r"^(\s*)#\s?", r"\1", ln) for ln in text.splitlines())   # un-comment
    m = re.search(r"df\.columns\s*=\s*(\[.*?\])", body, re.S)
    cols = ast.literal_eval(m.group(1)) if m else None
    p = re.search(r"df\.to_csv\(f?\"([^\"]+)\"\)", body)
    return cols, p.group(1) if p else None


def main():
    res = {}
    scripts = sorted(f for f in os.listdir(os.path.join(REF, "op_bm_scripts")) if f.startswith("benchmark_") and f.endswith(".py"))
    for f in scripts:
        text = open(os.path.join(REF, "op_bm_scripts", f)).read()
        op = re.search(r"op_name\s*=\s*\"([^\"]+)\"", text).group(1)
        cols, path = columns_and_path(text)
        if cols is None:   # benchmark_native_sort.py: rows go through the reference's DataWriter
            sys.path.insert(0, REF)
            sys.dont_write_bytecode = True
            from graph_benchmark.benchmark.DataWriter import DataWriter

            pn = re.search(r"param_names=\"([^\"]+)\"", text).group(1)
            dw = DataWriter(op_name=op, param_names=pn)
            dw.add_entry(params_lst=["1", "0", "True"], tshape=(3,), sparsity=0, bm_val=1.5)
            with tempfile.TemporaryDirectory() as tmp:
                dw.write_data(path=tmp)
                header = open(os.path.join(tmp, f"{op}.csv")).readline().rstrip("\n")
                first = open(os.path.join(tmp, f"{op}.csv")).readlines()[1].rstrip("\n")
            cols = next(csv.reader([header]))[1:]
            where = re.search(r"write_data\(path=os\.path\.join\(\"\./\", \"([^\"]+)\"\)\)", text).group(1)
            path = f"{where}/{op}.csv"
            res[op] = {"script": f, "csv": path, "columns": cols, "datawriter_row": first}
            continue
        res[op] = {"script": f, "csv": path.replace("{op_name}", op), "columns": cols}
    json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT, len(res), "scripts")


if __name__ == "__main__":
    main()
