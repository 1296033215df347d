#!/bin/bash
# round 5: s_memtime anatomy of single stages of the fused kernel (variants built with -DFDMI_SA_SUBSTAGE=k)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r5v
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for k in 11 10 2 5; do
  SUBSTAGE=$k FDMI_LIB=$PWD/foldingdiff_amd/_lib/sub$k/libfdmi.so timeout 200 python scripts/round5/sa_stamps.py 2>&1 | grep -A1 "it  [4-6]:" | head -6
done 2>&1 | tee gpurun_out/r5v/stamps.log
