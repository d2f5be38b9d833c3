"""BASELINE config 4 (F = 512, W = 256, out = 512, H = 150; 8 192 windows per call, chunked by the library): wall time and per-family
kernel times, with the chunks alternating between the two lanes (default) and on one lane (engine option "lanes" = 1).
usage: python profiles/config4_bench.py [windows] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mtad-gat-pytorch_amd")); sys.path.insert(0, ROOT)
import torch
if os.environ.get("VARIANT"):           # a developer build of the library (profiles/build_variants.sh) instead of the in-tree one
    import _native
    _native._LIB_PATH = os.path.join(ROOT, "profiles", "bin", "variants", os.environ["VARIANT"], "libmtadgat.so")
from mtad_gat import MTAD_GAT
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
kw = dict(n_features=512, window_size=256, out_dim=512, kernel_size=7, gru_hid_dim=150, forecast_n_layers=1, forecast_hid_dim=150, recon_hid_dim=150)
torch.manual_seed(0)
m = MTAD_GAT(**kw).to(dev).eval(); m.check_weight_contents = False
x = torch.rand(n, 256, 512, device=dev)
eng = m._sync_engine(dev)
chunks = [int(c) for c in os.environ.get("CHUNKS", "0").split(",")]
with torch.no_grad():
    m(x[:256])
    ref = None
    for chunk in chunks:
        if chunk: eng.set_chunk_windows(chunk)
        for lanes in [int(v) for v in os.environ.get("LANES", "0,1,0,1").split(",")]:
            eng.set_option("lanes", lanes)
            out = m(x); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps): out = m(x)
            torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
            eng.profile_enable(True); m(x); torch.cuda.synchronize(); prof = eng.profile_read(); eng.profile_enable(False)
            if ref is None: ref = out
            same = bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
            print(f"chunk {eng.chunk_windows():5d} lanes option {lanes} ({'two lanes' if lanes == 0 else 'one lane'}): {1e3 * t:8.2f} ms = {n / t:9.1f} windows/s   "
                  f"families {{{', '.join(f'{k}: {v[0]:.1f}' for k, v in prof.items() if v[1])}}} sum {sum(v[0] for v in prof.values()):.1f}  same {same}", flush=True)
    eng.set_option("lanes", 0)
