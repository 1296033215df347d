#!/bin/bash
# round 5: fused projection + attention kernel -- parity, per-kernel times, cycle stamps
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r5c
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python scripts/round5/sa_check.py > gpurun_out/r5c/sa_check.log 2>&1; echo "rc=$?" >> gpurun_out/r5c/sa_check.log
grep -v amdgpu.ids gpurun_out/r5c/sa_check.log | tail -12
{
for rep in 1 2; do
  TAG="c2 fused" FDMI_FUSE_ATTN=1 timeout 200 python scripts/kernel_times.py 2>&1 | tail -1
done
TAG="c2 two-kernel" FDMI_FUSE_ATTN=0 timeout 200 python scripts/kernel_times.py 2>&1 | tail -1
for v in ${VARIANTS:-}; do
  TAG="c2 fused $v" FDMI_FUSE_ATTN=1 FDMI_LIB=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so timeout 200 python scripts/kernel_times.py 2>&1 | tail -1
done
} | grep -v amdgpu.ids > gpurun_out/r5c/times.log
cat gpurun_out/r5c/times.log
timeout 300 python scripts/round5/sa_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5c/stamps.log
cat gpurun_out/r5c/stamps.log
