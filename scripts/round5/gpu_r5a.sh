#!/bin/bash
# round 5: same-box A/B of the q|k|v launch and the C5 step -- current library vs the round-3 V^T swizzle
# (-DFDMI_VT_SWZ_R3) vs the round-3 gemm_img.hip (git 388f932) in the current tree
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r5a
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
L=$PWD/foldingdiff_amd/_lib
{
for rep in 1 2 3; do
  for v in . vtr3 r3gemm; do
    TAG="c2 $v" FDMI_LIB=$L/$v/libfdmi.so timeout 200 python scripts/kernel_times.py 2>&1 | tail -1
  done
done
for rep in 1 2; do
  for v in . vtr3 r3gemm; do
    TAG="c5 $v" B=128 L=512 MAXPOS=512 FDMI_LIB=$L/$v/libfdmi.so timeout 200 python scripts/kernel_times.py 2>&1 | tail -1
  done
done
} | grep -v amdgpu.ids > gpurun_out/r5a/ab.log
cat gpurun_out/r5a/ab.log
