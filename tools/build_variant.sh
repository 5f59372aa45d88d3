#!/bin/bash
# Build a variant of the library for A/B timing (tools/ab.sh): tools/build_variant.sh <name> <file.hip> [-Dflags ...]
# Recompiles ONE translation unit with the extra flags and links it with the product build's other objects into
# amber_amd/csrc/variants/libamber_<name>.so (git-ignored, travels to the GPU box with the tree).
set -eu
name=$1; unit=$2; shift 2
cd "$(dirname "$0")/../amber_amd/csrc"
make -s -j8 >/dev/null
mkdir -p variants
extra=""
[ "$unit" = "mesh.hip" ] && extra="-ffp-contract=off"
./build_tu.sh "$unit" "variants/${name}_${unit%.hip}.o" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $extra "$@" | grep -v "^pkfix:" || true
objs=""
for o in api mpn_kernels mpn_f2 mpn_f2_general mpn_f2_general_emb mpn_bwd csr glue graph mesh; do
  if [ "$o" = "${unit%.hip}" ]; then objs="$objs variants/${name}_$o.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o "variants/libamber_$name.so"
echo "amber_amd/csrc/variants/libamber_$name.so"
