#!/bin/bash
# Builds csrc/libptmi_<tag>.so = the product library with extra -D flags on ONE kernel family, for same-process
# A/B timing (PT_LIBPTMI=<path> selects it in _abi.py).  Usage: tools/build_variant.sh <tag> <family .hip> <flags...>
set -e
TAG=$1; FAM=$2; shift 2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/g.p.u-pathtracer_amd"
make -j8 csrc/libptmi.so >/dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -include "$ROOT/tools/pt_exp_hooks.h" "$@" -c -o /tmp/${FAM%.hip}_$TAG.o csrc/$FAM
OBJS=""
for o in ptmi pt_build pt_k_mega pt_k_persist pt_k_wave; do
  if [ "$o.hip" = "$FAM" ]; then OBJS="$OBJS /tmp/${FAM%.hip}_$TAG.o"; else OBJS="$OBJS csrc/$o.o"; fi
done
$HIPCC --offload-arch=gfx950 -fPIC -shared -o csrc/libptmi_$TAG.so $OBJS
echo csrc/libptmi_$TAG.so
