#!/bin/bash
# SQ counter passes over the bench's eager launches (T = 3): matrix-pipe and VALU busy cycles, LDS conflicts, per kernel.
#   PMC_SETS="A B C;D E F" bash scripts/gpu_pmc.sh     (one rocprofv3 pass per ';' separated set; never with --stats / traces)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; cd /tmp; i=0
IFS=';' read -ra SETS <<< "${PMC_SETS:-SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM}"
for set in "${SETS[@]}"; do
  i=$((i+1)); echo "== rocprofv3 --pmc $set"
  FDMI_NO_GRAPH=1 timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$OUT/sq_$i -o pmc -- python $R/bench.py --steps 1 --warmup 0 --timesteps 3 --profile-every 0 --no-cpu-baseline --no-exact-f32 --no-c5-extra --no-user-paths > $R/$OUT/sq_$i.log 2>&1
  tail -1 $R/$OUT/sq_$i.log | cut -c1-160
done
cd $R
for j in $(seq 1 $i); do python scripts/pmc_summary.py $OUT/sq_$j 2>&1 | grep -E "gemm_img|attn_img|seq_attn|ffn16|embed_img|head_update_img|^#" | grep -v kernel_stats | cut -c1-170; done | tee $OUT/sq_summary.txt
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*.db" -delete
echo "== done"
