#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python scripts/stamps.py 2>&1 | grep -v amdgpu.ids | tee $OUT/stamps.log | head -150
R=$PWD; cd /tmp; i=0
IFS=';' read -ra SETS <<< "${PMC_EXTRA:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM}"
for set in "${SETS[@]}"; do
  i=$((i+1)); echo "== rocprofv3 --pmc $set"
  FDMI_NO_GRAPH=1 timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$OUT/pmcx_$i -o pmc -- python $R/bench.py --steps 1 --warmup 0 --timesteps 3 --profile-every 0 --no-cpu-baseline --no-exact-f32 > $R/$OUT/pmcx_$i.log 2>&1
  tail -1 $R/$OUT/pmcx_$i.log | cut -c1-160
done
cd $R
python scripts/pmc_summary.py $OUT/pmcx_1 2>&1 | grep -E "gemm_img|attn_img" | cut -c1-150
python scripts/pmc_summary.py $OUT/pmcx_2 2>&1 | grep -E "gemm_img|attn_img" | cut -c1-150
find $OUT -name "*kernel_trace.csv" -size +8M -delete
echo "== done"
