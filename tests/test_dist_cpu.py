"""N>1 path on CPU: world_size-2 `gloo` run of the Ulysses exchanges with the oracle attention — Wan (seq↔head all-to-all,
shard/gather: exact permutations, sharded forward == unsharded) and HunyuanVideo (joint image+text attention, latent / RoPE-table
split and gather).  Where /root/reference exists the same run also requires bit-equality with the reference's own exchange code
(comm/all2all.py, utils/wan/processor.py, ulysses/attn.py::ulysses_attn, utils/hunyuan/processor.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("world", [2, 3, 8])
def test_ulysses_gloo(world):
    """world 3: a non-power-of-two group — 6 heads, S = 96 + 1 padded to 99, and a Hunyuan token grid whose h axis (4) does not
    divide, so the split falls to the w axis.  world 8: the node size the scaling bench runs at — 16 heads, 12 rows per rank, the two-piece
    head->seq exchange with four destination ranks per piece (zero-row receive views on the other four)."""
    env = dict(os.environ, OMP_NUM_THREADS="1" if world > 4 else "2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(29531 + world),
           os.path.join(ROOT, "tests", "_dist_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "DIST_OK" in p.stdout
    sys.path.insert(0, ROOT)
    from oracle import ref_import

    if ref_import.reference_available():
        assert "REFERENCE_EXCHANGE_OK" in p.stdout and "REFERENCE_HUNYUAN_EXCHANGE_OK" in p.stdout
