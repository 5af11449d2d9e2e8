"""Builds libgut_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels to
the GPU box with the repository snapshot.  No JIT at import time: the product path fails loudly when
the library is missing (see _capi.py).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libgut_hip.so")
OBJ_DIR = os.path.join(HERE, "csrc", "_obj")

ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# (source, extra flags).  gut_project / gut_api carry the bit-exact numerics contract -> no FMA contraction.
UNITS = [
    ("gut_project.hip", ["-ffp-contract=off"]),
    ("gut_render.hip", ["-ffp-contract=fast", "-munsafe-fp-atomics", "-fno-slp-vectorize"]),
    ("gut_render_general.hip", ["-ffp-contract=fast", "-munsafe-fp-atomics", "-fno-slp-vectorize"]),
    ("gut_render_sorted.hip", ["-ffp-contract=fast", "-munsafe-fp-atomics", "-fno-slp-vectorize"]),
    ("gut_sort.hip", ["-Wno-unused-parameter"]),
    ("gut_ssim.hip", ["-ffp-contract=fast"]),
    ("gut_train.hip", ["-ffp-contract=fast"]),
    ("gut_api.cpp", ["-x", "hip", "-ffp-contract=off"]),
]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_diagnostic(out, defines, verbose=False):
    """A diagnostic build of the same sources with extra -D flags into `out` (its own object directory); tools only."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj_dir = out + ".obj"
    os.makedirs(obj_dir, exist_ok=True)
    objs = []
    for src, extra in UNITS:
        o = os.path.join(obj_dir, src + ".o")
        objs.append(o)
        cmd = [hipcc] + COMMON + extra + ["-D" + d for d in defines] + ["-c", os.path.join(CSRC, src), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    subprocess.check_call([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out] + objs)
    return out


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, "gut_internal.h"), os.path.join(CSRC, "gut_render_common.h"), os.path.join(CSRC, "gut_render.hip"),
               os.path.join(HERE, "..", "include", "gut_hip.h")]
    objs = []
    for src, extra in UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, src + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + COMMON + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
