"""Ulysses path on the GPU: two ranks share cuda:0 (one-GPU box), gloo process group with host-staged collectives
(test-only shim in tests/_dist_gpu_worker.py); the sharded HIP forward must match the single-GPU HIP forward."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 8])
def test_ulysses_world2_on_one_gpu(world):
    """world 8 = the node size, all eight ranks driving the one GPU of the box: the HIP Ulysses path (blocked exchange buffers, K-blocked output
    projection, sharded RoPE offsets with padding rows, the CFG branches on two streams) at the partitioning the scaling bench uses."""
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(29541 + 10 * (world > 2)),
           os.path.join(ROOT, "tests", "_dist_gpu_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "DIST_GPU_OK" in p.stdout


def test_ulysses_hunyuan_world2_on_one_gpu():
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29543",
           os.path.join(ROOT, "tests", "_dist_gpu_worker_hunyuan.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "DIST_GPU_HUNYUAN_OK" in p.stdout


def test_wan_vae_decode_dist_world2_on_one_gpu():
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29545",
           os.path.join(ROOT, "tests", "_dist_gpu_worker_vae.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "DIST_GPU_VAE_OK" in p.stdout


def test_hunyuan_vae_tile_parallel_world2_on_one_gpu():
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29547",
           os.path.join(ROOT, "tests", "_dist_gpu_worker_hunyuan_vae.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "DIST_GPU_HUNYUAN_VAE_OK" in p.stdout


def test_ulysses_over_rccl_world1():
    """The real RCCL backend on the box's one GPU (world size 1): same collectives, streams and events as the N-GPU run."""
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29549",
           os.path.join(ROOT, "tests", "_dist_gpu_worker_rccl1.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "DIST_GPU_RCCL1_OK" in p.stdout
