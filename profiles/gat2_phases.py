"""Cumulative time of the phases of k_gat2 (measurement hook "gat2_stop": the kernel returns after phase k; results invalid).
usage: python profiles/gat2_phases.py [windows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mtad-gat-pytorch_amd")); sys.path.insert(0, ROOT)
import torch
from bench import load_msl_state_dict
from mtad_gat import MTAD_GAT
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda", 0)
sd, kw = load_msl_state_dict()
model = MTAD_GAT(**kw); model.load_state_dict(sd); model = model.to(dev).eval(); model.check_weight_contents = False
x = torch.rand(n, kw["window_size"], kw["n_features"], generator=torch.Generator().manual_seed(1)).to(dev)
eng = model._sync_engine(dev)
names = {0: "full kernel", 1: "staging", 2: "+ projection", 3: "+ pair grid", 4: "+ reduce-scatter", 5: "+ VT build, softmax"}
with torch.no_grad():
    for gk in (1, 0):
        eng.set_option("gat_kernel", gk)
        for stop in ((0,) if gk == 1 else (1, 2, 3, 4, 5, 0)):
            eng.set_option("gat2_stop", stop)
            for _ in range(2): model(x)
            torch.cuda.synchronize(); eng.profile_enable(True)
            for _ in range(5): model(x)
            torch.cuda.synchronize(); prof = eng.profile_read(); eng.profile_enable(False)
            ms = sum(prof[k][0] for k in ("proj", "attend")) / 5
            print(f"{'k_gat (row-split)' if gk == 1 else 'k_gat2 ' + names[stop]:40s} both layers {ms:7.3f} ms per {n} windows")
    eng.set_option("gat2_stop", 0)
