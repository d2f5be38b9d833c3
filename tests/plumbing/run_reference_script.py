"""Launch one of the reference's own CLI scripts (train.py / predict.py) UNCHANGED, the way
SURVEY.md section 8c prescribes, with `import mtad_gat` resolved either to this package's drop-in
module ("ours") or to the reference's ("reference", the control run).

    python run_reference_script.py <ours|reference> <reference_root> <script.py> [script args ...]

The script runs with cwd = the caller's scratch directory (its data / output paths are relative).
Two import shims stand in for packages this image lacks -- neither touches the model path:
`more_itertools.consecutive_groups` (eval_methods.py:216) and `torch.utils.tensorboard.SummaryWriter`
(training.py:6; every run here passes --log_tensorboard False).
"""
import os
import runpy
import sys
import types

sys.dont_write_bytecode = True          # the reference tree is read-only


def _install_shims():
    try:
        import more_itertools  # noqa: F401
    except ModuleNotFoundError:
        mit = types.ModuleType("more_itertools")

        def consecutive_groups(iterable, ordering=lambda x: x):
            from itertools import groupby
            from operator import itemgetter
            for _, g in groupby(enumerate(iterable), key=lambda t: t[0] - ordering(t[1])):
                yield map(itemgetter(1), g)

        mit.consecutive_groups = consecutive_groups
        sys.modules["more_itertools"] = mit
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        tb = types.ModuleType("torch.utils.tensorboard")

        class SummaryWriter:                       # never instantiated: --log_tensorboard False
            def __init__(self, *a, **k):
                pass

            def add_text(self, *a, **k):
                pass

            def add_scalar(self, *a, **k):
                pass

        tb.SummaryWriter = SummaryWriter
        sys.modules["torch.utils.tensorboard"] = tb


def main():
    which, ref_root, script = sys.argv[1], sys.argv[2], sys.argv[3]
    here = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(os.path.dirname(os.path.dirname(here)), "mtad-gat-pytorch_amd")
    import matplotlib
    matplotlib.use("Agg")
    _install_shims()
    if which == "ours":
        sys.path.insert(0, pkg)                    # wins `from mtad_gat import MTAD_GAT` (train.py:7, predict.py:7)
        sys.path.insert(1, ref_root)               # utils, args, prediction, training, ...
    else:
        sys.path.insert(0, ref_root)
    seed = os.environ.get("PLUMBING_SEED")
    if seed is not None:
        import numpy as np
        import torch
        torch.manual_seed(int(seed))
        np.random.seed(int(seed))
    sys.argv = [script] + sys.argv[4:]
    runpy.run_path(os.path.join(ref_root, script), run_name="__main__")
    import mtad_gat
    print("MTAD_GAT_MODULE", os.path.abspath(mtad_gat.__file__))


if __name__ == "__main__":
    main()
