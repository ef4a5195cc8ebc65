"""One process per GPU: the launch contract shared by bench.py, tools/e2e.py and tools/hunyuan_bench.py.

`script --gpus N` arrives either from a launcher (`python -m torch.distributed.run --nproc-per-node N ... script --gpus N`: RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or bare (`python script --gpus N`).  The bare form re-runs the same command line
under torch.distributed.run on this node — rendezvous on 127.0.0.1 (the container hostname may not resolve) at a free port — and passes
the exit code through, so the first multi-GPU run needs no human to know the launcher incantation."""
import os
import socket
import subprocess
import sys


def self_launch(script_path, n):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what the host driver supports (RCCL / cross-process tensors)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(script_path), *sys.argv[1:]]
    print(f"{os.path.basename(script_path)}: no launcher environment, starting " + " ".join(cmd[1:8]) + " ...", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def ranks(script_path, gpus):
    """(world, rank, local_rank) for this process; re-launches (and exits with the job's code) when N > 1 was asked for without a launcher."""
    if gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(script_path, gpus))
    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world != gpus:
        raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    import torch

    if one_gpu_test():
        return world, rank, 0
    if torch.cuda.device_count() < local_rank + 1:
        raise SystemExit(f"{os.path.basename(script_path)}: rank {rank} needs cuda:{local_rank} but this node shows {torch.cuda.device_count()} GPU(s)")
    return world, rank, local_rank


def one_gpu_test():
    """X2V_ONE_GPU_TEST=1: a PLUMBING mode for boxes with a single GPU — every rank drives cuda:0, the process group is gloo and the two collective
    entry points the Ulysses driver uses stage device tensors through host memory (`host_staged_collectives`).  It exists so that the N-rank code
    paths of the drivers (self-check, branch streams, timers, JSON assembly) can be executed before a multi-GPU node ever sees them; its timings mean
    nothing and the drivers say so in their output."""
    return os.environ.get("X2V_ONE_GPU_TEST") == "1"


def host_staged_collectives(dist):
    """Wrap dist.all_to_all_single / dist.all_gather_into_tensor so that device tensors travel through host memory (gloo has no device path
    here).  Test plumbing for `one_gpu_test()` only."""
    import torch

    def staged(fn):
        def wrapped(out, inp, *a, **kw):
            if out.is_cuda:
                torch.cuda.current_stream().synchronize()
                o, i = torch.empty(out.shape, dtype=out.dtype), inp.detach().cpu()
                r = fn(o, i, *a, **kw)
                out.copy_(o)
                return r
            return fn(out, inp, *a, **kw)

        return wrapped

    if not getattr(dist.all_to_all_single, "_x2v_staged", False):
        dist.all_to_all_single = staged(dist.all_to_all_single)
        dist.all_to_all_single._x2v_staged = True
        dist.all_gather_into_tensor = staged(dist.all_gather_into_tensor)
