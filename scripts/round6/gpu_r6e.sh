#!/bin/bash
# round 6: ablations of the two-group kernel (wrong results by design): what bounds a projection stage / a slot
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r6e
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r6e
for v in . dbg1 dbg2 dbg3; do
  TAG="c2 fuse_attn=1 lib=$v" FDMI_FUSE_ATTN=1 FDMI_LIB=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 | sed "s/embed_ln_time.*head_update_wrap=[0-9.]* //"
done 2>&1 | tee $O/times.log
for v in dbg1 dbg2 dbg3; do
echo "== $v"; FDMI_LIB=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so timeout 300 python scripts/round6/sa16_stamps.py 2>&1 | grep -A14 "^wave 0\|^wave 4" | grep -v "slot 1[0-9]\|slot 2[0-9]"
done 2>&1 | tee $O/stamps.log
