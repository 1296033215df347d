#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python -m foldingdiff_amd.build 2>&1 | tail -1
{
python scripts/gemm_bench.py f16x3
FDMI_GEMM_PERSIST=0 python scripts/gemm_bench.py f16x3
for v in ${ABLATE_DBG:-}; do FDMI_GEMM_PERSIST=0 FDMI_GEMM_DBG=$v python scripts/gemm_bench.py f16x3; done
python scripts/gemm_bench.py f32
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_ablate.log
