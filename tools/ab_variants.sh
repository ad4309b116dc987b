#!/bin/bash
# Builds tuning variants of libzkmi355.so: tools/ab_variants.sh <tag> <file.hip> "<-D flags>"
# -> webauthn-halo2_amd/build/libzkmi355_<tag>.so (run with ZKMI355_LIB=<that path>)
set -e
cd "$(dirname "$0")/../webauthn-halo2_amd"
tag=$1; src=$2; defs=$3
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result"
base=$(basename $src .hip)
hipcc $FLAGS $defs -c csrc/$src -o build/${base}_$tag.o
objs=""
for s in engine ntt msm poly prover_kernels quotient prover serde; do
  if [ $s = $base ]; then objs="$objs build/${base}_$tag.o"; else objs="$objs build/$s.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o build/libzkmi355_$tag.so $objs
echo "built build/libzkmi355_$tag.so"
