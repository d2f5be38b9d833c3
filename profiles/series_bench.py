"""score_series over a long series (MSL shape): ms per call, windows/s.   python profiles/series_bench.py [rows]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mtad-gat-pytorch_amd")); sys.path.insert(0, ROOT)
import torch
from mtad_gat import MTAD_GAT
dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536 + 100
kw = dict(n_features=55, window_size=100, out_dim=1, kernel_size=7, gru_hid_dim=150, forecast_n_layers=3, forecast_hid_dim=150,
          recon_hid_dim=150, dropout=0.3, alpha=0.2)
torch.manual_seed(0)
m = MTAD_GAT(**kw).to(dev).eval()
m.check_weight_contents = False
series = torch.rand(rows, 55, device=dev)
with torch.no_grad():
    for _ in range(2):
        m.score_series(series)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        m.score_series(series)
    torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
prof = ""
try:
    eng = m._engine
    eng.profile_enable(True)
    with torch.no_grad():
        m.score_series(series)
    torch.cuda.synchronize()
    prof = str(eng.profile_read())
except Exception as e:
    prof = repr(e)
print(f"rows={rows}  score_series {ms:.3f} ms  {(rows - 100) / ms / 1e3:.3f} M windows/s  {prof}", flush=True)
