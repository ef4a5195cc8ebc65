#!/bin/bash
# Builds a copy of libx2v_hip.so with extra compiler flags into tools/probes/ab/<tag>/ for A/B runs on the GPU box:
#   tools/build_variant.sh trail8 -DX2V_G256_TRAIL=8        then      LD_LIBRARY_PATH=tools/probes/ab/trail8 tools/x2v_check pgemm ...
# ONLY=<sources> (space separated, without .hip) recompiles just those with the flags and links the other objects of the main build
# (lightx2v_amd/csrc/build/, `python -m lightx2v_amd.build` first):   ONLY="attn" tools/build_variant.sh rot -DSOME_SWITCH=1
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
out=tools/probes/ab/$tag
mkdir -p $out/obj
ALL="x2v_api norm gemm gemm256 gemm256s gemm256c gemm256c8 attn quant_fp8 conv3d vae vae16g mx sched probe"
SRC=${ONLY:-$ALL}
for s in $SRC; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I include -I lightx2v_amd/csrc "$@" -c lightx2v_amd/csrc/$s.hip -o $out/obj/$s.o &
done
wait
for s in $ALL; do
  [ -f $out/obj/$s.o ] || cp lightx2v_amd/csrc/build/$s.o $out/obj/$s.o || { echo "no object for $s.hip"; exit 1; }
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libx2v_hip.so $out/obj/*.o
rm -rf $out/obj
ls -la $out/libx2v_hip.so
