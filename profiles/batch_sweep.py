"""Throughput of the eval forward over the batch size (MSL shape, fp32 results, check_weight_contents = False):
    python profiles/batch_sweep.py [batch ...]      -> one line per batch size"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mtad-gat-pytorch_amd")); sys.path.insert(0, ROOT)
import torch
from mtad_gat import MTAD_GAT
dev = torch.device("cuda:0")
sizes = [int(a) for a in sys.argv[1:]] or [256, 1024, 2048, 4096, 8192, 12288, 16384, 32768, 65536, 131072]
kw = dict(n_features=55, window_size=100, out_dim=1, kernel_size=7, gru_hid_dim=150, forecast_n_layers=3, forecast_hid_dim=150,
          recon_hid_dim=150, dropout=0.3, alpha=0.2)
torch.manual_seed(0)
m = MTAD_GAT(**kw).to(dev).eval()
m.check_weight_contents = False
for B in sizes:
    x = torch.rand(B, 100, 55, device=dev)
    n = max(3, min(50, int(4e6 / B / 10)))
    with torch.no_grad():
        for _ in range(2):
            m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            m(x)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"B={B:7d}  {ms:8.3f} ms/forward  {B / ms / 1e3:6.3f} M windows/s", flush=True)
    del x
