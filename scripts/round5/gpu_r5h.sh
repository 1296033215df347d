#!/bin/bash
# round 5: full GPU test suite + a short bench with the fused kernel as the default
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r5h
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5h/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r5h/pytest.log
tail -5 gpurun_out/r5h/pytest.log
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32 > gpurun_out/r5h/bench.log 2>&1
tail -1 gpurun_out/r5h/bench.log | cut -c1-1500
