"""Time the weight-stationary GEMM of every library variant under foldingdiff_amd/_lib/<variant>/ (ablation builds, FDMI_WS_DBG).
   python scripts/ws_time.py [variant ...]"""
import ctypes as C
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(65536, 768, 384), (65536, 1280, 384), (65536, 256, 384)]

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        sys.path.insert(0, ROOT)
        from foldingdiff_amd import _binding
        lib = _binding.load()
        out = []
        for (M, N, K) in SHAPES:
            ms = C.c_double()
            _binding.check(lib.fd_test_gemm_time(0, _binding.FD_PREC["f16x3"], M, N, K, 20, C.byref(ms)))
            out.append(f"{N}: {ms.value * 1e3:7.1f}")
        print("  ".join(out), flush=True)
        sys.exit(0)
    variants = sys.argv[1:] or ["", *sorted(os.path.basename(os.path.dirname(p)) for p in glob.glob(f"{ROOT}/foldingdiff_amd/_lib/*/libfdmi.so"))]
    for v in variants:
        env = dict(os.environ, FDMI_GEMM_WS="1")
        if v:
            env["FDMI_LIB"] = f"{ROOT}/foldingdiff_amd/_lib/{v}/libfdmi.so"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        print(f"{v or 'product':10s} us by N  {r.stdout.strip()}  {r.stderr.strip()[-200:] if r.returncode else ''}", flush=True)
        if os.environ.get("FDMI_WS_STAMPS"):
            print("\n".join(l for l in r.stderr.splitlines() if "ws stamps" in l or "wave" in l), flush=True)
