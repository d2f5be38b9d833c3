"""Phase timeline of k_gath from in-kernel s_memtime stamps (developer build -DMTADGAT_GATH_STAMP of mtadgat_gath.hip:
TU=mtadgat_gath bash profiles/build_variants.sh stamp "-DMTADGAT_GATH_STAMP").  64 mid-launch workgroups per layer record the shader
clock at every phase boundary, per wave; this prints, per layer, the mean time of each phase (last wave to arrive), the spread
between the waves, and which sampled workgroups shared a CU.
usage: python profiles/gath_timeline.py [windows] [variant]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mtad-gat-pytorch_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
import _native
variant = sys.argv[2] if len(sys.argv) > 2 else "stamp"
_native._LIB_PATH = os.path.join(ROOT, "profiles", "bin", "variants", variant, "libmtadgat.so")
from bench import load_msl_state_dict
from mtad_gat import MTAD_GAT
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda", 0)
sd, kw = load_msl_state_dict()
model = MTAD_GAT(**kw); model.load_state_dict(sd); model = model.to(dev).eval(); model.check_weight_contents = False
x = torch.rand(n, kw["window_size"], kw["n_features"], generator=torch.Generator().manual_seed(1)).to(dev)
with torch.no_grad():
    for _ in range(3): model(x)
torch.cuda.synchronize()
lib = _native.load_library()
NW, NS, WINS = 8, 32, 64
buf = (ctypes.c_uint64 * (2 * WINS * NW * NS))()
lib.mtadgat_debug_gath_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.mtadgat_debug_gath_stamps(buf, len(buf))
assert rc == 0, rc
st = np.frombuffer(buf, dtype=np.uint64).reshape(2, WINS, NW, NS).astype(np.int64)
names = {0: "start", 1: "x loaded + max", 2: "scale known (2 barriers)", 3: "pieces in LDS (barrier)", 4: "conv MFMA loop done", 5: "conv epilogue done",
         6: "barrier + range flag", 7: "fill + barrier: parts begin", 23: "c/d read + barrier", 24: "softmax done", 25: "aggregation done",
         26: "out tile written", 27: "barrier", 28: "rows stored: end"}
for p in range(5):
    names[8 + 3 * p] = f"part {p}: projection done"
    names[9 + 3 * p] = f"part {p}: barrier (pair grid begins)"
    names[10 + 3 * p] = f"part {p}: pair grid done"
for layer, lname in ((0, "temporal (+conv)"), (1, "feature")):
    s = st[layer]
    ok = s[:, 0, 0] > 0
    s = s[ok]
    if len(s) == 0:
        print(lname, "no stamps"); continue
    t0 = s[:, :, 0].min(axis=1)                       # workgroup start = first wave's start
    print(f"== {lname}: {len(s)} workgroups; cycles since the workgroup's first wave started (mean over workgroups)")
    print(f"{'id':>3} {'phase':40s} {'first wave':>10} {'last wave':>10} {'delta(last)':>11}")
    prev = 0.0
    ids = [i for i in sorted(names) if (s[:, :, i] > 0).any()]
    for i in ids:
        v = s[:, :, i].astype(np.float64)
        v[v == 0] = np.nan
        rel = v - t0[:, None]
        first = np.nanmean(np.nanmin(rel, axis=1)); last = np.nanmean(np.nanmax(rel, axis=1))
        print(f"{i:3d} {names[i]:40s} {first:10.0f} {last:10.0f} {last - prev:11.0f}")
        prev = last
    tot = (s[:, :, 28].max(axis=1) - t0)
    print(f"   total per window: mean {tot.mean():.0f} cycles, min {tot.min()}, max {tot.max()}")
    hw = s[:, 0, 31]
    cu = ((hw >> 32) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
    order = np.argsort(t0)
    print("   sampled workgroups by start time: (start - first start, duration, cu tag, simd of wave 0)")
    for k in order[:24]:
        print(f"     +{t0[k] - t0.min():9d} {tot[k]:8d}  cu {cu[k]:06x} simd {(hw[k] >> 4) & 3}")
