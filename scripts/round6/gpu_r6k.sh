#!/bin/bash
# round 6: SQ counters of seq_attn16 (lockstep, pipelined projection): matrix pipe busy, LDS conflicts / activity
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r6k
export FDMI_FUSE_ATTN=1
bash scripts/gpu_pmc.sh 2>&1 | grep -v "^==" | tail -3
grep -E "seq_attn16" gpurun_out/sq_summary.txt 2>/dev/null | tee gpurun_out/r6k/sq_seq_attn16.txt
for j in 1 2; do python scripts/pmc_summary.py gpurun_out/sq_$j 2>&1 | grep -E "seq_attn16" | cut -c1-150; done | tee gpurun_out/r6k/sq_seq_attn16.txt
