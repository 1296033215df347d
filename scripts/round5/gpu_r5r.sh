#!/bin/bash
# round 5: the embed kernel's sensitivity to its grid (per-workgroup LDS parameter fill) and lanes per row -- no code change
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r5r
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for lpt in 8 16; do for g in 256 512 1024 2048 4096; do
  TAG="grid=$g lpt=$lpt" FDMI_ROW_GRID=$g FDMI_ROW_LPT=$lpt timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 | sed "s/gemm_attn_out.*gemm_head_dense1=[0-9.]* //"
done; done 2>&1 | tee gpurun_out/r5r/sweep.log
