"""Training step at BASELINE config 4's shape (F=512, W=256, out=512, H=150; 256 windows, eval-mode dropout 0): forward + RMSE
losses + HIP backward + Adam on the wide attention kernels.   python profiles/train_step_c4.py [batch] [steps]
(under rocprofv3 --kernel-trace --stats for the per-kernel table)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mtad-gat-pytorch_amd")); sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from mtad_gat import MTAD_GAT
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
kw = dict(n_features=512, window_size=256, out_dim=512, kernel_size=7, gru_hid_dim=150, forecast_n_layers=1, forecast_hid_dim=150, recon_hid_dim=150)
torch.manual_seed(0)
m = MTAD_GAT(**kw).to(dev).train()
m.check_weight_contents = False
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
x = torch.rand(B, 256, 512, device=dev); y = torch.rand(B, 512, device=dev)
def step():
    opt.zero_grad()
    p, r = m(x)
    (torch.sqrt(F.mse_loss(y, p)) + torch.sqrt(F.mse_loss(x, r))).backward()
    opt.step()
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): step()
torch.cuda.synchronize()
print(f"config 4 training step, {B} windows: {1e3 * (time.perf_counter() - t0) / N:.2f} ms  grad_path {getattr(m, 'grad_path', None)}", flush=True)
