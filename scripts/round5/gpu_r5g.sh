#!/bin/bash
# round 5: A/B of fused-kernel variants: bitwise check (first case only) + per-kernel times
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r5g
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in ${VARIANTS:-.}; do
  lib=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so
  FDMI_LIB=$lib timeout 300 python scripts/round5/sa_check.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/[$v] /"
  for rep in 1 2; do
    TAG="c2 fused $v" FDMI_FUSE_ATTN=1 FDMI_LIB=$lib timeout 200 python scripts/kernel_times.py 2>&1 | tail -1
  done
done > gpurun_out/r5g/ab.log 2>&1
TAG="c2 two-kernel" FDMI_FUSE_ATTN=0 timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 >> gpurun_out/r5g/ab.log
cat gpurun_out/r5g/ab.log
