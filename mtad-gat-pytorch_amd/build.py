"""Build libmtadgat.so (gfx950) in-tree with hipcc.  No torch C++ extension, no JIT cache:
the .so sits next to this file so it travels with the source tree to the GPU box."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["mtadgat_kernels.hip", "mtadgat_attend.hip", "mtadgat_convw.hip", "mtadgat_gat.hip", "mtadgat_gath.hip", "mtadgat_gru.hip", "mtadgat_gru_f32.hip", "mtadgat_gru_bf16.hip", "mtadgat_gru_x3.hip", "mtadgat_gru_x3b.hip", "mtadgat_gru_cm.hip", "mtadgat_gru16.hip", "mtadgat_packdev.hip", "mtadgat_bwd.hip", "mtadgat_bwdw.hip", "mtadgat_eval.hip", "mtadgat_pack.cpp", "mtadgat_capi.cpp"]
HEADERS = ["mtadgat_kernels.h", "mtadgat_device.h", "mtadgat_host.h", "mtadgat_gru_impl.h", "mtadgat_gat_impl.h", os.path.join("..", "..", "include", "mtadgat.h")]
LIB = os.path.join(HERE, "libmtadgat.so")
OBJDIR = os.path.join(HERE, "build")          # object files (git-ignored)
# -fno-slp-vectorize: keep the attention inner loops as single-issue v_add_f32 (2 VALU/element);
# SLP packing into v_pk_add_f32 costs 3 VALU/element there.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile the HIP kernels + C ABI into libmtadgat.so; returns the library path.
    One hipcc per stale translation unit (source or any header newer than its object), run concurrently,
    then one link step.  (Developer variants of single translation units: profiles/build_variants.sh.)"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = []
    stamp = os.path.join(OBJDIR, ".flags")
    flags_now = " ".join(FLAGS + extra)
    flags_same = (not os.path.isdir(OBJDIR) and not extra) or (os.path.exists(stamp) and open(stamp).read() == flags_now)
    if not force and not _stale() and flags_same:
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    hdr_t = max(hdr_t, os.path.getmtime(os.path.abspath(__file__)))
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        srcp = os.path.join(CSRC, src)
        if not force and flags_same and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(srcp), hdr_t):
            continue
        cmd = [hipcc] + FLAGS + extra + ["-x", "hip", "-c", srcp, "-o", obj]
        if verbose:
            print("[mtadgat] " + " ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    open(stamp, "w").write(flags_now)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB + ".tmp"]
    if verbose:
        print("[mtadgat] " + " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
