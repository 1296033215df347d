"""
Build libfdmi.so (the gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m foldingdiff_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU present.  The shared object lands in
foldingdiff_amd/_lib/ (git-ignored, shipped to the GPU box with the tree).
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "_lib")
LIB_PATH = os.path.join(LIB_DIR, "libfdmi.so")
SOURCES = ["api.hip", "gemm_f32.hip", "gemm_img.hip", "attention_f32.hip", "attention_img.hip", "rowwise.hip",
           "rowwise_img.hip", "nerf.hip"]
HEADERS = [os.path.join(CSRC, "fdmi_kernels.h"), os.path.join(CSRC, "img_common.h"),
           os.path.join(os.path.dirname(PKG_DIR), "include", "fdmi.h")]
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def find_hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 and link libfdmi.so.  Returns its path."""
    hipcc = find_hipcc()
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(op)
        if force or _stale(op, [sp] + HEADERS):
            jobs.append([hipcc] + CXXFLAGS + ["-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed ({r.returncode}):\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB_PATH, objs):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH])
    return LIB_PATH


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print(p)
