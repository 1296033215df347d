#!/bin/bash
# round 5: first run of the fused projection + attention kernel -- parity (bitwise against the two-kernel path), then timing;
# and the C5 half of the r3 -> r4 A/B that gpu_r5a.sh lost to a shell variable clash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r5b
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
LIBD=$PWD/foldingdiff_amd/_lib
timeout 600 python scripts/round5/sa_check.py > gpurun_out/r5b/sa_check.log 2>&1; echo "rc=$?" >> gpurun_out/r5b/sa_check.log
grep -v amdgpu.ids gpurun_out/r5b/sa_check.log | tail -15
{
for rep in 1 2; do
  TAG="c2 fused" FDMI_FUSE_ATTN=1 timeout 200 python scripts/kernel_times.py 2>&1 | tail -1
  TAG="c2 two-kernel" FDMI_FUSE_ATTN=0 timeout 200 python scripts/kernel_times.py 2>&1 | tail -1
done
for v in . vtr3; do
  TAG="c5 $v" B=128 L=512 MAXPOS=512 FDMI_LIB=$LIBD/$v/libfdmi.so timeout 200 python scripts/kernel_times.py 2>&1 | tail -1
done
} | grep -v amdgpu.ids > gpurun_out/r5b/times.log
cat gpurun_out/r5b/times.log
