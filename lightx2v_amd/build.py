"""Builds libx2v_hip.so (the C-ABI kernel library) for gfx950 with hipcc — in-tree, no cmake/JIT cache,
so the .so travels with the repo snapshot to the GPU box.  `python -m lightx2v_amd.build [--force]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OBJ_DIR = os.path.join(CSRC, "build")
LIB_PATH = os.path.join(HERE, "libx2v_hip.so")
SOURCES = ["x2v_api.hip", "norm.hip", "gemm.hip", "gemm256.hip", "gemm256s.hip", "gemm256c.hip", "gemm256c8.hip", "attn.hip", "quant_fp8.hip", "conv3d.hip", "vae.hip", "vae16g.hip", "mx.hip", "sched.hip", "probe.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-I", INCLUDE, "-I", CSRC]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    deps = [os.path.join(CSRC, src), os.path.join(CSRC, "x2v_common.h"), os.path.join(INCLUDE, "x2v.h")]
    if _stale(obj, deps):
        subprocess.run([HIPCC, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj], check=True)
    return obj


def build(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if force or _stale(LIB_PATH, objs):
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs], check=True)
    if verbose:
        print("built", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
