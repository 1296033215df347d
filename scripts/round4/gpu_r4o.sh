#!/bin/bash
# round 4: attention with the S^T tiles of the next position multiplied at the end of the current one -- parity, same-box A/B
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r4o
L=$PWD/foldingdiff_amd/_lib
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "forward or step or loop or varlen or packed or relative or long or c3 or c5 or mask" > gpurun_out/r4o/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r4o/pytest.log
tail -4 gpurun_out/r4o/pytest.log
{
for rep in 1 2; do
TAG=snext1 timeout 200 python scripts/c3_times.py
TAG=snext0 FDMI_LIB=$L/snext0/libfdmi.so timeout 200 python scripts/c3_times.py
done
} 2>&1 | grep -v amdgpu.ids | grep "chunk\|c2:\|c5:" | sed "s/embed_ln_time=[0-9.]* gemm_qkv=[0-9.]* //; s/gemm_attn_out.*| /| /" > gpurun_out/r4o/ab.log
cat gpurun_out/r4o/ab.log
