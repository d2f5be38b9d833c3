"""Eval forward of a small batch (the reference Predictor's 256 windows, prediction.py:31; MSL shape):
    python profiles/forward_small.py <batch>
used under rocprofv3 --kernel-trace --stats by profiles/collect.sh."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mtad-gat-pytorch_amd")); sys.path.insert(0, ROOT)
import torch
if os.environ.get("VARIANT"):           # a developer build of the library (profiles/build_variants.sh) instead of the in-tree one
    import _native
    _native._LIB_PATH = os.path.join(ROOT, "profiles", "bin", "variants", os.environ["VARIANT"], "libmtadgat.so")
from mtad_gat import MTAD_GAT
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kw = dict(n_features=55, window_size=100, out_dim=1, kernel_size=7, gru_hid_dim=150, forecast_n_layers=3, forecast_hid_dim=150,
          recon_hid_dim=150, dropout=0.3, alpha=0.2)
torch.manual_seed(0)
m = MTAD_GAT(**kw).to(dev).eval()
x = torch.rand(B, 100, 55, device=dev)
N = 50
with torch.no_grad():
    for _ in range(5):
        m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N):
        m(x)
    torch.cuda.synchronize()
print(f"B={B} ms/forward {(time.perf_counter() - t0) / N * 1e3:.3f} ({N + 5} forwards in the trace)", flush=True)
