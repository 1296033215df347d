#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r5e
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python scripts/round5/sa_ctx_diag.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5e/diag.log
cat gpurun_out/r5e/diag.log
