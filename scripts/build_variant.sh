#!/bin/bash
# Build an experimental engine under exp/ for scripts/exp_variants.sh:  scripts/build_variant.sh NAME [-DMACRO=VALUE ...] [SRC_ROOT]
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root
defs=()
for a in "$@"; do case "$a" in -D*) defs+=("$a");; *) src=$a;; esac; done
mkdir -p $root/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-function \
  "${defs[@]}" -I$src/include -o $root/exp/$name.so $src/advntr_amd/csrc/engine.hip
