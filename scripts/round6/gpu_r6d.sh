#!/bin/bash
# round 6: two-group seq_attn16.hip with ILP-structured attention pieces: parity, launch times (default and wave-priority variant), slot anatomy
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r6d
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r6d
CASES=0,1,2,3,5,7 timeout 900 python scripts/round6/sa16_check.py 2>&1 | tail -14 | tee $O/sa16_check.log
for rep in 1 2; do
for v in . prio2; do
  TAG="c2 fuse_attn=1 lib=$v" FDMI_FUSE_ATTN=1 FDMI_LIB=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so timeout 200 python scripts/kernel_times.py 2>&1 | tail -1
done
TAG="c2 fuse_attn=2" FDMI_FUSE_ATTN=2 timeout 200 python scripts/kernel_times.py 2>&1 | tail -1
done 2>&1 | tee $O/times.log
timeout 300 python scripts/round6/sa16_stamps.py 2>&1 | tail -140 | grep -v "^  slot [12][0-9]" | tee $O/stamps.log
