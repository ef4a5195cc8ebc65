#!/bin/bash
# Builds libx2v_hip.so PLUS the producer / consumer attention probe (tools/probes/attn_pc.hip, entry x2v_probe_attn_pc) into tools/probes/ab/<tag>/:
#   tools/probes/build_attn_pc.sh pc [-DX2V_PC_KNOCK=4 ...]     then     LD_LIBRARY_PATH=tools/probes/ab/pc tools/x2v_check pattn 13 75600 40 6
# (the other objects come from the main build, lightx2v_amd/csrc/build/: `python -m lightx2v_amd.build` first)
set -e
cd "$(dirname "$0")/../.."
tag=$1; shift
out=tools/probes/ab/$tag
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I include -I lightx2v_amd/csrc "$@" -c tools/probes/attn_pc.hip -o $out/attn_pc.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libx2v_hip.so lightx2v_amd/csrc/build/*.o $out/attn_pc.o
rm -f $out/attn_pc.o
ls -la $out/libx2v_hip.so
