"""Knock-out timing of k_gath (measurement hook "gath_dbg" as a bit mask: 1 no pair grid, 2 no projection, 4 return before the
softmax; results invalid).  usage: python profiles/gath_knockout.py [windows]"""
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
names = {0: "full kernel", 1: "no pair grid", 2: "no projection", 3: "no pair grid, no projection", 4: "no softmax / aggregation / output",
         5: "staging + projection only", 6: "staging + pair grid only", 7: "staging only"}
with torch.no_grad():
    for mask in (0, 1, 2, 3, 4, 5, 6, 7):
        eng.set_option("gath_dbg", mask)
        for _ in range(2): model(x)
        torch.cuda.synchronize(); eng.profile_enable(True)
        for _ in range(5): model(x)
        torch.cuda.synchronize(); prof = eng.profile_read(); eng.profile_enable(False)
        ms = sum(prof[k][0] for k in ("proj", "attend")) / 5
        print(f"k_gath {names[mask]:40s} both layers {ms:7.3f} ms per {n} windows")
    eng.set_option("gath_dbg", 0)
