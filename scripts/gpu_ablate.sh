#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python -m foldingdiff_amd.build 2>&1 | tail -1
{
python scripts/gemm_bench.py f16x3
FDMI_GEMM_DBG=1 python scripts/gemm_bench.py f16x3
FDMI_GEMM_DBG=2 python scripts/gemm_bench.py f16x3
FDMI_GEMM_DBG=3 python scripts/gemm_bench.py f16x3
FDMI_GEMM_BM=128 python scripts/gemm_bench.py f16x3
python scripts/gemm_bench.py f32
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_ablate.log
