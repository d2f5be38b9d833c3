"""A/B of k_gath measurement options (engine option "gath_dbg") on the flagship workload: both attention layers' time per call.
usage: python profiles/gath_ab.py <windows> <reps> <dbg> [<dbg> ...]   (environment switches, e.g. MTADGAT_GATH_WL=1, apply to the process)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mtad-gat-pytorch_amd")); sys.path.insert(0, ROOT)
import torch
from bench import load_msl_state_dict
from mtad_gat import MTAD_GAT
n = int(sys.argv[1]); reps = int(sys.argv[2]); vals = [int(v) for v in sys.argv[3:]] or [0]
dev = torch.device("cuda", 0)
sd, kw = load_msl_state_dict()
model = MTAD_GAT(**kw); model.load_state_dict(sd); model = model.to(dev).eval(); model.check_weight_contents = False
x = torch.rand(n, kw["window_size"], kw["n_features"], generator=torch.Generator().manual_seed(1)).to(dev)
eng = model._sync_engine(dev)
ref = None
with torch.no_grad():
    for rnd in range(2):
        for v in vals:
            eng.set_option("gath_dbg", v)
            for _ in range(2): out = model(x)
            torch.cuda.synchronize(); eng.profile_enable(True)
            for _ in range(reps): out = model(x)
            torch.cuda.synchronize(); prof = eng.profile_read(); eng.profile_enable(False)
            if ref is None: ref = out
            same = bool(torch.equal(out[1], ref[1]))
            ms = {k: round(prof[k][0] / reps, 3) for k in prof if prof[k][1]}
            print(f"gath_dbg {v:3d}  attention {ms.get('attend', 0) + ms.get('proj', 0):7.3f} ms   all {ms}  same_as_first {same}", flush=True)
    eng.set_option("gath_dbg", 0)
