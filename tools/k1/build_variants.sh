#!/bin/bash
# Build container: variant builds of libat3hip.so for same-box A/B on the GPU box (they travel with the snapshot; build_ab/ is git-ignored).
# usage: tools/k1/build_variants.sh "name|-DFLAG ..." "name2|..."      -> build_ab/lib_<name>.so
cd "$(dirname "$0")/../.."
C=atracdenc_amd/csrc
mkdir -p build_ab
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -fPIC"
for f in at1hip.hip at3phip.hip at3_tables.cpp; do
  o=build_ab/${f%.*}.o
  if [ ! -f $o ] || [ $C/$f -nt $o ]; then hipcc $FL -c $C/$f -o $o & fi
done
wait
build_one() {
  local name=${1%%|*} flags=${1#*|}
  [ "$flags" = "$1" ] && flags=""
  hipcc $FL $flags -c $C/at3hip.hip -o build_ab/at3hip_$name.o 2> build_ab/$name.log || { echo "BUILD FAILED $name"; tail -5 build_ab/$name.log; return 1; }
  hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/exports.map -o build_ab/lib_$name.so build_ab/at3hip_$name.o build_ab/at1hip.o build_ab/at3phip.o build_ab/at3_tables.o
  echo "built build_ab/lib_$name.so  [$flags]"
}
N=0
for v in "$@"; do
  build_one "$v" &
  N=$((N + 1)); [ $((N % 6)) -eq 0 ] && wait
done
wait
