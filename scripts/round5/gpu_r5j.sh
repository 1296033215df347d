#!/bin/bash
# round 5: timing-only A/B of (wrong-result) ablation builds of the fused kernel
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r5j
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1; do
for v in ${VARIANTS:-.}; do
  TAG="c2 fused $v" FDMI_FUSE_ATTN=1 FDMI_LIB=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 | sed "s/embed_ln_time.*head_update_wrap=[0-9.]* //"
done
done > gpurun_out/r5j/ab.log 2>&1
cat gpurun_out/r5j/ab.log
