"""
Build libfdmi.so (the gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m foldingdiff_amd.build [--force] [variant DEFINE[=VALUE] | @source=-flag | rev:source.hip=REV ...]

An experiment argument of the form ``@gemm_img=-fno-slp-vectorize`` adds a compiler flag for that one source
(``@all=...`` for every source); ``rev:seq_attn.hip=HEAD~3`` takes that one source from a git revision (everything
else from the working tree): the "before" leg of a same-box A/B.

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
SOURCES = ["api.hip", "gemm_f32.hip", "gemm_img.hip", "gemm_ws.hip", "gemm_ln_rows.hip", "attention_f32.hip", "attention_img.hip", "attention_gen.hip", "seq_attn.hip", "seq_attn16.hip", "ffn16.hip", "rowwise.hip",
           "rowwise_img.hip", "nerf.hip"]
HEADERS = [os.path.join(CSRC, "fdmi_kernels.h"), os.path.join(CSRC, "img_common.h"),
           os.path.join(os.path.dirname(PKG_DIR), "include", "fdmi.h")]
ARCH = "gfx950"
# flags of single sources (see the header of the source for the reason)
PER_SOURCE_FLAGS = {
    # packed fp32 VALU instructions (v_pk_fma_f32 ...) serialize with the matrix pipe, plain ones issue beside another wave's
    # MFMAs (profiles/r03_coissue2_probe.log): the staggered attention schedule needs un-packed softmax / skew arithmetic
    "attention_img": ["-fno-slp-vectorize"],
    "attention_gen": ["-fno-slp-vectorize"],
    "seq_attn": ["-fno-slp-vectorize"],
    "seq_attn16": ["-fno-slp-vectorize"],
    "ffn16": ["-fno-slp-vectorize"],
}
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


def build(force: bool = False, verbose: bool = False, variant: str = "", defines=()) -> str:
    """Compile every HIP source for gfx950 and link libfdmi.so.  Returns its path.
    ``variant`` + ``defines``: an experiment build with extra -D flags in _lib/<variant>/libfdmi.so, selected at run
    time with FDMI_LIB=<path> (same-box A/B measurements; the default library is untouched)."""
    hipcc = find_hipcc()
    os.makedirs(LIB_DIR, exist_ok=True)
    lib_dir = os.path.join(LIB_DIR, variant) if variant else LIB_DIR
    lib_path = os.path.join(lib_dir, "libfdmi.so")
    obj_dir = os.path.join(lib_dir, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    jobs = []
    objs = []
    revs = dict(d[4:].split("=", 1) for d in defines if d.startswith("rev:"))
    defines = [d for d in defines if not d.startswith("rev:")]
    old_sources = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if src in revs:   # the old revision of this source, beside the current ones so that its includes resolve
            sp = os.path.join(CSRC, f"_rev_{variant}_{src}")
            r = subprocess.run(["git", "show", f"{revs[src]}:foldingdiff_amd/csrc/{src}"], cwd=PKG_DIR, capture_output=True, text=True, check=True)
            with open(sp, "w") as fh:
                fh.write(r.stdout)
            old_sources.append(sp)
            force = True
        op = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(op)
        if force or _stale(op, [sp] + HEADERS):
            stem = src.replace(".hip", "")
            extra = [d.split("=", 1)[1] for d in defines if d.startswith("@") and d[1:].split("=", 1)[0] in (stem, "all")]
            jobs.append([hipcc] + CXXFLAGS + PER_SOURCE_FLAGS.get(stem, []) + extra
                        + [f"-D{d}" for d in defines if not d.startswith("@")] + ["-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed ({r.returncode}):\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    try:
        with ThreadPoolExecutor(max_workers=8) as ex:
            list(ex.map(run, jobs))
        counts = mfma_counts(defines, {src: os.path.join(CSRC, f"_rev_{variant}_{src}") for src in revs}) if variant else None
    finally:
        for sp in old_sources:
            os.remove(sp)
    if force or jobs or _stale(lib_path, objs):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC"] + objs + ["-o", lib_path])
    if variant:
        # an experiment build says what it is: its -D flags, and the MFMA count of every kernel of the sources those flags can reach,
        # next to the default build's (scripts/kernel_times.py refuses to time a library whose counts differ unless told that it is an
        # ablation: round 5 read a dead-code-eliminated build as "97 % of the matrix peak")
        import json
        meta = {"defines": list(defines), "revs": revs, "mfma": counts, "mfma_default": mfma_counts(())}
        with open(os.path.join(lib_dir, "build_info.json"), "w") as fh:
            json.dump(meta, fh, indent=1, sort_keys=True)
    return lib_path


MFMA_SOURCES = ["seq_attn16.hip", "ffn16.hip", "seq_attn.hip", "gemm_img.hip", "attention_img.hip"]


def mfma_counts(defines=(), paths=None) -> dict:
    """{kernel symbol: number of v_mfma instructions in its gfx950 code} for the MFMA kernels, compiled with `defines`
    (hipcc -S, device side only; a dead-code-eliminated ablation shows up as a lower count)."""
    import re
    import tempfile
    hipcc = find_hipcc()
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for src in MFMA_SOURCES:
            stem = src.replace(".hip", "")
            extra = [d.split("=", 1)[1] for d in defines if d.startswith("@") and d[1:].split("=", 1)[0] in (stem, "all")]
            asm = os.path.join(td, stem + ".s")
            cmd = [hipcc, "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-Wno-unused-function"] + PER_SOURCE_FLAGS.get(stem, []) + extra \
                + [f"-D{d}" for d in defines if not d.startswith("@")] + ["-S", "--cuda-device-only", "-o", asm, (paths or {}).get(src, os.path.join(CSRC, src))]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc -S failed for {src}:\n{r.stderr[-2000:]}")
            text = open(asm).read()
            for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
                n = m.group(2).count("v_mfma")
                if n:
                    out[m.group(1)] = n
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]   # [variant [DEFINE[=VALUE] ...]]
    p = build(force="--force" in sys.argv, verbose=True, variant=args[0] if args else "", defines=args[1:])
    print(p)
