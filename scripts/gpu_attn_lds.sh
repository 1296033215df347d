#!/bin/bash
# LDS bank-conflict counter of the attention kernel for library variants (ablation builds ad<N>: FDMI_ATTN_DBG=<N>)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; cd /tmp
: > $R/$OUT/attn_lds.log
for v in ${VARIANTS:-. ad1 ad2 ad4 ad8}; do
  lib=$R/foldingdiff_amd/_lib/$v/libfdmi.so
  rm -rf $R/$OUT/al_$v
  FDMI_LIB=$lib FDMI_NO_GRAPH=1 timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/$OUT/al_$v -o pmc -- python $R/bench.py --steps 1 --warmup 0 --timesteps 2 --profile-every 0 --no-cpu-baseline --no-exact-f32 --no-c5-extra --no-user-paths > $R/$OUT/al_$v.log 2>&1
  (cd $R; python scripts/pmc_summary.py $OUT/al_$v 2>&1 | grep -E "attn_img" | grep -v "calls=" | sed "s/^/[$v] /" | cut -c1-150) | tee -a $R/$OUT/attn_lds.log
  find $R/$OUT/al_$v -name "*.csv" -delete
done
echo "== done"
