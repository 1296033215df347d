#!/bin/bash
# Build an A/B library variant in which ONE source comes from an older git revision (everything else: the working tree):
#   bash scripts/build_rev_variant.sh <variant> <rev> <source.hip> [-DFLAG ...]     -> foldingdiff_amd/_lib/<variant>/libfdmi.so
set -eu
cd "$(dirname "$0")/.."
V=$1; REV=$2; SRC=$3; shift 3
TMP=$(mktemp -d)
cp -r foldingdiff_amd/csrc $TMP/csrc
mkdir -p $TMP/include && cp include/fdmi.h $TMP/include/
git show "$REV:foldingdiff_amd/csrc/$SRC" > $TMP/csrc/$SRC
OUT=foldingdiff_amd/_lib/$V; mkdir -p $OUT/obj
# api.hip includes ../../include/fdmi.h relative to csrc
mkdir -p $TMP/x/y && mv $TMP/csrc $TMP/x/y/csrc && mv $TMP/include $TMP/x/include
pids=()
for f in api gemm_f32 gemm_img gemm_ws gemm_ln_rows attention_f32 attention_img attention_gen seq_attn rowwise rowwise_img nerf; do
  extra=""; { [ "$f" = attention_img ] || [ "$f" = attention_gen ] || [ "$f" = seq_attn ]; } && extra="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $extra "$@" -c $TMP/x/y/csrc/$f.hip -o $OUT/obj/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/obj/*.o -o $OUT/libfdmi.so
rm -rf $TMP
echo $OUT/libfdmi.so
