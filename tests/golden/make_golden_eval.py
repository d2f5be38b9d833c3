#!/usr/bin/env python3
"""Known-answer fixture for the device evaluation kernels ("post-processing is pinned"):
the anomaly scores of the shipped MSL run (reference output/MSL/27062021_111641/{train,test}_output.pkl), its labels
and the numbers its summary.txt holds.  Run where /root/reference exists:

    python -B tests/golden/make_golden_eval.py
"""
import json
import os

import numpy as np
import pandas as pd

REF = os.environ.get("MTAD_REFERENCE", "/root/reference")
D = os.path.join(REF, "output", "MSL", "27062021_111641")
HERE = os.path.dirname(os.path.abspath(__file__))

tr = pd.read_pickle(os.path.join(D, "train_output.pkl"))
te = pd.read_pickle(os.path.join(D, "test_output.pkl"))
summary = json.load(open(os.path.join(D, "summary.txt")))
n = 20000
out = dict(train_scores=tr["A_Score_Global"].values.astype(np.float32), test_scores=te["A_Score_Global"].values.astype(np.float32),
           test_labels=te["A_True_Global"].values.astype(np.uint8),
           forecast=te["Forecast_0"].values[:n].astype(np.float32), recon=te["Recon_0"].values[:n].astype(np.float32),
           true=te["True_0"].values[:n].astype(np.float32), a_score_0=te["A_Score_0"].values[:n].astype(np.float32),
           thresh_0=np.float64(te["Thresh_0"].values[0]), train_score_0=tr["A_Score_0"].values.astype(np.float32),
           summary=np.frombuffer(json.dumps(summary).encode(), dtype=np.uint8))
path = os.path.join(HERE, "msl_eval.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path) / 1e6, "MB", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
