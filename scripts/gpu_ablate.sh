#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python -m foldingdiff_amd.build 2>&1 | tail -1
{
for v in ${ABLATE_VARS:-256 128}; do FDMI_GEMM_PBM=$v python scripts/gemm_bench.py f16x3; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_ablate3.log
