"""Training step (BASELINE config 3 shape: SMD F=38, W=100, out=38, dropout 0.3, Adam) phase by phase:
    python profiles/train_step.py <batch>
used under rocprofv3 --kernel-trace --stats by profiles/collect.sh."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mtad-gat-pytorch_amd")); sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from mtad_gat import MTAD_GAT
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kw = dict(n_features=38, window_size=100, out_dim=38, kernel_size=7, gru_hid_dim=150, forecast_n_layers=3, forecast_hid_dim=150, recon_hid_dim=150, dropout=0.3, alpha=0.2)
torch.manual_seed(0)
m = MTAD_GAT(**kw).to(dev).train()
m.check_weight_contents = False
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
g = torch.Generator().manual_seed(1)
x = torch.rand(B, 100, 38, generator=g).to(dev); y = torch.rand(B, 38, generator=g).to(dev)
def sync(): torch.cuda.synchronize()
T = {"sync_engine": 0, "fwd": 0, "loss": 0, "bwd": 0, "opt": 0}
N = 20
for it in range(N + 3):
    if it == 3:
        for k in T: T[k] = 0
    sync(); t0 = time.perf_counter()
    opt.zero_grad()
    eng = m._sync_engine(dev); sync(); t1 = time.perf_counter()
    p, r = m(x); sync(); t2 = time.perf_counter()
    loss = torch.sqrt(F.mse_loss(y, p)) + torch.sqrt(F.mse_loss(x, r)); sync(); t3 = time.perf_counter()
    loss.backward(); sync(); t4 = time.perf_counter()
    opt.step(); sync(); t5 = time.perf_counter()
    T["sync_engine"] += t1 - t0; T["fwd"] += t2 - t1; T["loss"] += t3 - t2; T["bwd"] += t4 - t3; T["opt"] += t5 - t4
print(f"B={B}", {k: round(1e3 * v / N, 3) for k, v in T.items()}, "total ms", round(1e3 * sum(T.values()) / N, 3), flush=True)
