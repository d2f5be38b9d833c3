"""A/B of one engine option on the flagship forward (MSL shape): python profiles/ab_option.py <option> <value_a> <value_b> [windows] [reps]
Prints ms per forward and the per-family event times for both settings, interleaved (a b a b ...) to average out clock drift."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mtad-gat-pytorch_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import load_msl_state_dict  # noqa: E402
from mtad_gat import MTAD_GAT  # noqa: E402


def main():
    opt, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    dev = torch.device("cuda:0")
    sd, kw = load_msl_state_dict()
    model = MTAD_GAT(**kw)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    model.precision = "fp32"
    model.check_weight_contents = False
    x = torch.rand(n, kw["window_size"], kw["n_features"], generator=torch.Generator().manual_seed(1234)).to(dev)
    eng = model._sync_engine(dev)
    res = {va: [], vb: []}
    fam = {va: None, vb: None}
    with torch.no_grad():
        for v in (va, vb):
            eng.set_option(opt, v)
            for _ in range(2):
                model(x)
        torch.cuda.synchronize(dev)
        for _ in range(reps):
            for v in (va, vb):
                eng.set_option(opt, v)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(5):
                    model(x)
                torch.cuda.synchronize(dev)
                res[v].append((time.perf_counter() - t0) / 5 * 1e3)
        for v in (va, vb):
            eng.set_option(opt, v)
            eng.profile_enable(True)
            for _ in range(3):
                model(x)
            torch.cuda.synchronize(dev)
            fam[v] = {k: round(t[0] / 3, 3) for k, t in eng.profile_read().items() if t[1]}
            eng.profile_enable(False)
        eng.set_option(opt, va)
    for v in (va, vb):
        print(f"{opt}={v}: ms per forward of {n} windows: {' '.join(f'{t:.3f}' for t in res[v])}  min {min(res[v]):.3f}  families {fam[v]}")


if __name__ == "__main__":
    main()
