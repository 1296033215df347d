"""Row-count sweep, tile kernel (gemm_img.hip) against the weight-stationary kernel (gemm_ws.hip), plain-bias epilogue, K = 384.
   python scripts/ws_sweep.py"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MS = [512, 1024, 2048, 4096, 8192, 12288, 16384, 24576, 32768, 49152, 65536]
NS = [384, 768, 1152]

if __name__ == "__main__":
    if len(sys.argv) > 1:
        sys.path.insert(0, ROOT)
        from foldingdiff_amd import _binding
        lib = _binding.load()
        for N in NS:
            out = []
            for M in MS:
                ms = C.c_double()
                _binding.check(lib.fd_test_gemm_time(0, _binding.FD_PREC["f16x3"], M, N, 384, 30, C.byref(ms)))
                out.append(f"{ms.value * 1e3:6.1f}")
            print(f"{sys.argv[1]:5s} N={N:5d} " + " ".join(out), flush=True)
        sys.exit(0)
    print("rows            " + " ".join(f"{m:6d}" for m in MS))
    for tag, v in (("tile", "0"), ("ws", "1")):
        subprocess.run([sys.executable, os.path.abspath(__file__), tag], env=dict(os.environ, FDMI_GEMM_WS=v), check=True)
