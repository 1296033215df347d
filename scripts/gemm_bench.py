#!/usr/bin/env python3
"""Micro-benchmark of the token GEMM kernels (C-ABI hook fd_test_gemm_time).
   python scripts/gemm_bench.py [f32|f16x3] ; env FDMI_GEMM_* select experiment variants."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foldingdiff_amd import _binding
lib = _binding.load()
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("FDMI_GEMM"))
for (M, N, K) in ((65536, 1152, 384), (65536, 384, 384), (65536, 768, 384), (65536, 384, 768)):
    ms = C.c_double()
    _binding.check(lib.fd_test_gemm_time(0, _binding.FD_PREC[prec], M, N, K, 20, C.byref(ms)))
    print(f"{prec:6s} [{tag}] M={M} N={N} K={K}: {ms.value*1e3:8.1f} us  {2.0*M*N*K/ms.value/1e9:7.1f} TFLOP/s", flush=True)
