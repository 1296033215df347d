#!/bin/bash
# round 4: the weight-stationary GEMM -- parity in the hook tests, bit identity and time against the tile kernel
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r4j
timeout 300 python scripts/ws_check.py > gpurun_out/r4j/ws_check.log 2>&1; echo "ws_check rc=$?" >> gpurun_out/r4j/ws_check.log
FDMI_GEMM_WS=1 timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "gemm_kernel_vs_fp64" > gpurun_out/r4j/pytest_ws.log 2>&1
tail -3 gpurun_out/r4j/pytest_ws.log; cat gpurun_out/r4j/ws_check.log
