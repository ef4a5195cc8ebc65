#!/bin/bash
# A/B build for the attention kernel generations (VERDICT r3 #3-iii): the product tree carries only the 16x16x32 kernel (v9); the
# 32x32x16 ping-pong body (v8) was deleted in round 3.  This script builds tools/probes/ab/attn_r3/libx2v_hip.so = today's library with
# csrc/attn.hip taken from commit 21aed1c (the last one that held BOTH bodies behind X2V_ATTN_GEN=8|9, same launcher rules and flags) plus a shim for
# the one export that commit did not have yet.  Nothing of it ships:
#     X2V_LIB_PATH=$PWD/tools/probes/ab/attn_r3/libx2v_hip.so X2V_ATTN_GEN=8 python bench.py --steps 20 --warmup 5 --no-cpu-baseline     (v8)
#     X2V_LIB_PATH=$PWD/tools/probes/ab/attn_r3/libx2v_hip.so X2V_ATTN_GEN=9 python bench.py --steps 20 --warmup 5 --no-cpu-baseline     (v9 of that commit)
set -e
cd "$(dirname "$0")/.."
COMMIT=${ATTN_COMMIT:-21aed1c}
out=tools/probes/ab/attn_r3
mkdir -p $out/obj $out/src
git show $COMMIT:lightx2v_amd/csrc/attn.hip > $out/src/attn.hip
cat > $out/src/attn_shim.hip <<'EOS'
#include "x2v_common.h"
// x2v_attn_vt_launch_plan did not exist at the commit the attention source of this A/B build comes from; lib.py only needs the symbol to load.
extern "C" __attribute__((visibility("default"))) int x2v_attn_vt_launch_plan(int64_t, int64_t, int, int, int) { return X2V_E_ARG; }
EOS
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I include -I lightx2v_amd/csrc"
for s in x2v_api norm gemm gemm256 gemm256s gemm256c quant_fp8 conv3d vae mx sched probe; do
  /opt/rocm/bin/hipcc $FLAGS -c lightx2v_amd/csrc/$s.hip -o $out/obj/$s.o &
done
/opt/rocm/bin/hipcc $FLAGS -c $out/src/attn.hip -o $out/obj/attn.o &
/opt/rocm/bin/hipcc $FLAGS -c $out/src/attn_shim.hip -o $out/obj/attn_shim.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libx2v_hip.so $out/obj/*.o
rm -rf $out/obj $out/src
ls -la $out/libx2v_hip.so
