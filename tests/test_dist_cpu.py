"""N>1 path on CPU: world_size-2 `gloo` run of the Ulysses exchange (seq↔head all-to-all, shard/gather) with the
oracle attention — must reproduce the single-process result exactly."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ulysses_world2_gloo():
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "tests", "_dist_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "DIST_OK" in p.stdout
