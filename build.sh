#!/bin/bash
# Builds webauthn-halo2_amd/libzkmi355.so for gfx950 (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")/webauthn-halo2_amd"
OUT=libzkmi355.so
SRCS="csrc/engine.hip csrc/ntt.hip csrc/msm.hip csrc/poly.hip csrc/prover_kernels.hip csrc/quotient.hip csrc/prover.hip csrc/serde.hip"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result"
mkdir -p build
objs=""
pids=""
for s in $SRCS; do
  o=build/$(basename $s .hip).o
  objs="$objs $o"
  if [ ! -f $o ] || [ $s -nt $o ] || [ -n "$(find csrc ../include -name '*.h' -newer $o)" ]; then
    hipcc $FLAGS -c $s -o $o &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $objs
echo "built $(pwd)/$OUT"
# the C++ host example of the same ABI (examples/prove_host.cpp): plain g++, linked against the library above
cd ..
g++ -O2 -std=c++17 -Wall -Iinclude examples/prove_host.cpp -Lwebauthn-halo2_amd -lzkmi355 -Wl,-rpath,'$ORIGIN/../webauthn-halo2_amd' -o examples/prove_host
echo "built $(pwd)/examples/prove_host"
# the phase-level host (examples/prove_host_phases.cpp): its host-side field / transcript code comes from the engine's host headers,
# which carry HIP's function attributes — hence hipcc; it links against the same library through the public ABI only
hipcc --offload-arch=gfx950 -O2 -std=c++17 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -x hip examples/prove_host_phases.cpp -Lwebauthn-halo2_amd -lzkmi355 -Wl,-rpath,'$ORIGIN/../webauthn-halo2_amd' -o examples/prove_host_phases
echo "built $(pwd)/examples/prove_host_phases"
