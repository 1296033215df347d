#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r6n
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r6n
for rep in 1 2; do
for v in . prio2; do
  TAG="c2 fuse_attn=1 lib=$v" FDMI_FUSE_ATTN=1 FDMI_LIB=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 | sed "s/embed_ln_time.*head_update_wrap=[0-9.]* //"
done
TAG="c2 fuse_attn=2" FDMI_FUSE_ATTN=2 timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 | sed "s/embed_ln_time.*head_update_wrap=[0-9.]* //"
done 2>&1 | tee $O/times.log
FDMI_LIB=$PWD/foldingdiff_amd/_lib/prio2/libfdmi.so timeout 300 python scripts/round6/sa16_stamps.py 2>&1 | grep "mean\|^wave" | tee $O/stamps.log
