#!/bin/bash
# round 4: few-rows LayerNorm GEMM -- LN hook tests, model-level bit identity, small-batch step times
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r4n
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "few_rows or gemm_layernorm or ws_bit" > gpurun_out/r4n/pytest_ln.log 2>&1; echo "rc=$?" >> gpurun_out/r4n/pytest_ln.log
tail -15 gpurun_out/r4n/pytest_ln.log
{
TAG=auto timeout 200 python scripts/small_batch_times.py
TAG=ln16k FDMI_LN_ROWS_MAX=16384 timeout 200 python scripts/small_batch_times.py
TAG=tile FDMI_GEMM_WS=0 timeout 200 python scripts/small_batch_times.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r4n/small_batch.log
cat gpurun_out/r4n/small_batch.log
