#!/bin/bash
# Same-box A/B on ONE gpurun box: the only kind of comparison the profiles/ notes draw conclusions from (box-to-box spread is 3-6 %).
#
#   LEGS="<lib>[:ENV=VAL[,ENV=VAL...]] ..."   the legs, run alternately; <lib> = "." (default library) or a variant of
#                                             `python -m foldingdiff_amd.build <variant> DEFINE... | rev:source.hip=REV`
#   REPS=2            alternations
#   BURST=1           per-kernel microseconds of a few reverse steps, eager launches (scripts/kernel_times.py)
#   SUSTAINED=1       bench.py --steps 1 per leg: backbones/s, ms per timestep, the fused kernel's sustained launch time
#   C3=1              per-chunk kernel times of the C3 length sweep (scripts/c3_times.py)
#   CHECK="<pytest -k expression>"   parity tests first, on the default library (a leg that fails parity is not worth timing)
#   CHECK_LEGS=1      ... and on every leg's library as well (FDMI_LIB)
#   STAMPS=<script>   a cycle-stamp script (scripts/stamps.py, scripts/sa16_stamps.py) per leg, after the timings
#   ALLOW_ABLATION=1  let kernel_times.py time libraries whose MFMA counts differ from the default build's (it prints the counts)
#   OUT=gpurun_out/ab
#
# e.g. the round-6 comparison of the two fused kernels:
#   gpurun --timeout 1500 -- 'LEGS=".:FDMI_FUSE_ATTN=1 .:FDMI_FUSE_ATTN=2 .:FDMI_FUSE_ATTN=0" SUSTAINED=1 bash scripts/gpu_ab.sh'
set -u
cd "$(dirname "$0")/.."
OUT=${OUT:-gpurun_out/ab}; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
LEGS=${LEGS:-.}
abl=""; [ "${ALLOW_ABLATION:-0}" = 1 ] && abl="--allow-ablation"

leg_env() {   # "<lib>:A=1,B=2" -> env assignments on stdout
  local lib=${1%%:*} rest=""
  [[ "$1" == *:* ]] && rest=${1#*:}
  [ "$lib" != "." ] && echo "FDMI_LIB=$PWD/foldingdiff_amd/_lib/$lib/libfdmi.so"
  echo "$rest" | tr ',' '\n' | grep = || true
}

if [ -n "${CHECK:-}" ]; then
  timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "$CHECK" 2>&1 | tail -4 | sed "s/^/[check .] /" | tee $OUT/check.log
  if [ "${CHECK_LEGS:-0}" = 1 ]; then
    for leg in $LEGS; do
      env $(leg_env "$leg") timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "$CHECK" 2>&1 | tail -4 | sed "s/^/[check $leg] /" | tee -a $OUT/check.log
    done
  fi
fi
: > $OUT/burst.log; : > $OUT/sustained.log; : > $OUT/c3.log
for rep in $(seq 1 ${REPS:-2}); do
  if [ "${BURST:-1}" = 1 ]; then
    for leg in $LEGS; do
      env $(leg_env "$leg") TAG="[$leg]" timeout 300 python scripts/kernel_times.py $abl 2>&1 | tail -1 | tee -a $OUT/burst.log
    done
  fi
  if [ "${SUSTAINED:-0}" = 1 ]; then
    for leg in $LEGS; do
      env $(leg_env "$leg") timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-c5-extra --no-user-paths --no-traffic 2>&1 | tail -1 \
        | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('[$leg]', 'backbones/s', round(d['value'],2), 'ms/timestep', round(d['ms_per_step']/1000,3), ' '.join(f'{n}={v[\"avg_ms\"]*1e3:.1f}' for n,v in k.items() if 'attention' in n or 'qkv' in n))" \
        | tee -a $OUT/sustained.log
    done
  fi
  if [ "${C3:-0}" = 1 ]; then
    for leg in $LEGS; do
      env $(leg_env "$leg") TAG="[$leg]" timeout 400 python scripts/c3_times.py 2>&1 | grep "chunk\|c2:\|b8" | cut -c1-330 | tee -a $OUT/c3.log
    done
  fi
done
if [ -n "${STAMPS:-}" ]; then
  for leg in $LEGS; do
    echo "== [$leg]"; env $(leg_env "$leg") timeout 300 python $STAMPS 2>&1 | grep -v amdgpu.ids
  done > $OUT/stamps.log
fi
echo "== done"
