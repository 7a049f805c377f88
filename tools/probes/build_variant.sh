#!/bin/bash
# builds tools/probes/libaz_mcts_<tag>.so from the working tree with extra compiler flags (narrow instantiation set)
# usage: tools/probes/build_variant.sh <tag> [-DFLAG ...]
set -e
tag=$1; shift
cd "$(dirname "$0")/../../alphazero_amd/csrc"
out=build_var_$tag
mkdir -p $out
FLAGS="--offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -fPIC -fno-fast-math -Wno-unused-function -Wno-pass-failed -DAZ_NARROW_BUILD $*"
for f in az_api az_search az_fused az_replay az_minatar; do
  /opt/rocm/bin/hipcc $FLAGS -c -o $out/$f.o $f.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probes/libaz_mcts_$tag.so $out/*.o
rm -rf $out
