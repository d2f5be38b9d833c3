import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import mtad_gat_oracle as oracle
z = np.load(os.path.join(ROOT, "tests", "golden", "msl.npz"))
sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
print("cpu_count", os.cpu_count(), "default threads", torch.get_num_threads(), flush=True)
for nt in (8, 32, 64, 128):
    torch.set_num_threads(nt)
    for b in (256,):
        x = torch.rand(b, 100, 55)
        with torch.no_grad():
            t0 = time.perf_counter(); oracle.forward(x, sd, 0.2, aten_gru=True); t1 = time.perf_counter()
            oracle.forward(x, sd, 0.2, aten_gru=True); t2 = time.perf_counter()
        print(f"threads={nt} b={b}: first {t1-t0:.2f}s second {t2-t1:.2f}s -> {b/(t2-t1):.1f} windows/s", flush=True)
