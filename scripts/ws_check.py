"""Weight-stationary GEMM (gemm_ws.hip) against the tile kernel (gemm_img.hip): bit identity and time.  Each variant runs in its own
process because the switch (FDMI_GEMM_WS) is read once.   python scripts/ws_check.py"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

SHAPES = [(65536, 768, 384), (65536, 1152, 384), (65536, 1280, 384), (65536, 256, 384), (34036, 768, 384), (1000, 768, 384)]


def child(tag):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from foldingdiff_amd import _binding
    lib = _binding.load()
    P = C.c_void_p
    for (M, N, K) in SHAPES:
        ms = C.c_double()
        _binding.check(lib.fd_test_gemm_time(0, _binding.FD_PREC["f16x3"], M, N, K, 20, C.byref(ms)))
        print(f"{tag} time {M}x{N}x{K}: {ms.value * 1e3:8.1f} us", flush=True)
    for epi in (0, 1):
        for (M, N, K) in [(1000, 768, 384), (300, 1152, 384), (4096 + 37, 1280, 384), (256, 128, 384), (9000, 96, 384)]:
            rng = np.random.default_rng(M + N + epi)
            A = rng.standard_normal((M, K)).astype(np.float32)
            A[rng.random((M, K)) < 0.3] *= 1e-4
            W = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
            b = (rng.standard_normal(N) * 0.05).astype(np.float32)
            R = np.zeros((M, N), dtype=np.float32)
            Cc = np.empty((M, N), dtype=np.float32)
            _binding.check(lib.fd_test_gemm(0, _binding.FD_PREC["f16x3"], epi, A.ctypes.data_as(P), W.ctypes.data_as(P), b.ctypes.data_as(P),
                                            R.ctypes.data_as(P), Cc.ctypes.data_as(P), M, N, K))
            np.save(f"/tmp/ws_{tag}_{epi}_{M}_{N}.npy", Cc)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    for tag, v in (("tile", "0"), ("ws", "1")):
        env = dict(os.environ, FDMI_GEMM_WS=v)
        subprocess.run([sys.executable, os.path.abspath(__file__), tag], env=env, check=True)
    import glob
    bad = 0
    for f in sorted(glob.glob("/tmp/ws_tile_*.npy")):
        a, b = np.load(f), np.load(f.replace("ws_tile_", "ws_ws_"))
        same = np.array_equal(a, b)
        bad += not same
        print(os.path.basename(f), "bit-identical" if same else f"DIFFERENT max {np.abs(a - b).max():.3e} nan {np.isnan(b).sum()}")
    sys.exit(1 if bad else 0)
