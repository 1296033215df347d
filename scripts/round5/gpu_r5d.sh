#!/bin/bash
# round 5: cycle stamps of ablation variants of the fused kernel
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r5d
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in ${VARIANTS:-sa_noattn}; do
  echo "== $v"
  FDMI_LIB=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so timeout 300 python scripts/round5/sa_stamps.py 2>&1 | grep -v amdgpu.ids | head -${NLINES:-30}
done > gpurun_out/r5d/stamps.log
cat gpurun_out/r5d/stamps.log
