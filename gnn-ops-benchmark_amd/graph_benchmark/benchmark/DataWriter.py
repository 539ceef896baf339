"""CSV row collector used by benchmark_native_sort.py (reference:
graph_benchmark/benchmark/DataWriter.py:5-36). Same constructor, add_param_names / add_entry /
write_data and the same four-column layout, so CSVs stay comparable with the reference's data/*.csv.
"""
import os

import pandas as pd

_FIXED_COLUMNS = ("Input size (>95% mem util)*", "Sparsity", "GPU clock time")


class DataWriter:
    def __init__(self, op_name, param_names=None):
        self._rows = []
        self._op_name = op_name
        self._param_names = param_names

    def add_param_names(self, param_names):
        self._param_names = param_names

    def add_entry(self, params_lst, tshape, sparsity, bm_val, delimiter=";"):
        self._rows.append([delimiter.join(params_lst), str(tshape), sparsity, bm_val])

    def write_data(self, path=None):
        frame = pd.DataFrame(self._rows, columns=[self._param_names, *_FIXED_COLUMNS])
        if path is not None:
            frame.to_csv(os.path.join(path, f"{self._op_name}.csv"))
        return frame
