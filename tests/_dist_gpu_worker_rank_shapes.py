"""world_size-8 worker of tests/test_gpu_rank_shapes.py: ONE transformer block at the real dimensions of BASELINE config #3 (Wan2.1-14B, 75 600
tokens, 40 heads -> 9 450 tokens and 5 heads per rank) or #5 (HunyuanVideo-13B, 118 800 + 256 tokens, 24 heads -> 14 850 image tokens and 3 heads
per rank) through the product's Ulysses driver, all eight ranks on the box's one GPU (gloo with host-staged collectives: tests/_dist_gpu_worker.py).
The parent test wrote the block-boundary inputs to $X2V_RANK_SHAPES_DIR/inputs.pt; every rank writes the sampled rows of its shard to
out_rank<r>.pt and the parent compares them with the CPU oracle."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _dist_gpu_worker import _host_staged  # noqa: E402


def wan(r, n, d, out_dir):
    from lightx2v_amd import scheduler, synth, wan as W

    dims = dict(synth.WAN_DIMS["wan2.1-14b"], num_layers=1)
    wl = synth.WORKLOADS["wan14b_720px81f"]
    ts = wl["target_shape"]
    wd = synth.synth_wan_weights(dims, seed=31)  # the parent's weights (seeded)
    cfg = W.default_config(dims, target_shape=ts, target_video_length=wl["frames"], infer_steps=4, parallel_attn_type="ulysses")
    model = W.WanModel(cfg, {k: v.cuda() for k, v in wd.items()})
    del wd
    sch = scheduler.WanScheduler(cfg, device="cuda")
    model.set_scheduler(sch)
    tr = model.transformer_infer
    assert tr.sp_world == n and tr.sp_rank == r
    S = d["x"].shape[0]
    s_local = S // n
    x = d["x"][r * s_local : (r + 1) * s_local].cuda()
    grid_sizes = torch.tensor([d["grid"]], dtype=torch.long)
    rope = W.rope_cos_sin_table(128, "cuda")
    out = tr.infer_block(model.transformer_weights.blocks[0], grid_sizes, d["embed"].cuda(), x, d["embed0"].cuda(), torch.tensor([S]), rope, d["context"].cuda())
    torch.cuda.synchronize()
    pa = tr.parallel_attention
    assert pa.copies == 0 and pa._buffers, "the fused driver must take the copy-free blocked exchange path"
    b = next(iter(pa._buffers.values()))
    assert tuple(b["sq"].shape) == (n, s_local, 40 * 128 // n) == (8, 9450, 640), tuple(b["sq"].shape)
    assert pa.split_head2seq is True, "two-piece head->seq (the designed path)"
    assert torch.isfinite(out.float()).all()
    torch.save(out[d["local_rows"].cuda()].cpu(), os.path.join(out_dir, f"out_rank{r}.pt"))
    return "RANK_SHAPES_WAN_OK"


def hunyuan(r, n, d, out_dir):
    from lightx2v_amd import hunyuan as hy, synth, ulysses

    dims = dict(synth.HUNYUAN_DIMS["hunyuan-13b"], double_blocks=1, single_blocks=0)
    wd = {k: v for k, v in synth.synth_hunyuan_weights(dict(dims, double_blocks=1, single_blocks=1), seed=21).items() if k.startswith("double_blocks.")}
    cfg = hy.default_config(dims, infer_steps=4)
    tw = hy.HunyuanTransformerWeights(cfg)
    tw.load({k: v.cuda() for k, v in wd.items()})
    tr = hy.HunyuanTransformerInfer(cfg)
    tr.parallel_attention = ulysses.UlyssesHunyuanAttention()
    gt, gh, gw = d["grid"]
    w_local = gw // n
    D = d["img"].shape[1]
    sl = slice(r * w_local, (r + 1) * w_local)
    # utils/hunyuan/processor.py:5-50: the rank's slab of the token grid (W is the axis 8 divides) and the RoPE rows of the same tokens
    img = d["img"].view(gt, gh, gw, D)[:, :, sl].reshape(-1, D).contiguous().cuda()
    cos = d["cos"].view(gt, gh, gw, -1)[:, :, sl].reshape(img.shape[0], -1).contiguous().cuda()
    sin = d["sin"].view(gt, gh, gw, -1)[:, :, sl].reshape(img.shape[0], -1).contiguous().cuda()
    n_local, n_txt = img.shape[0], d["txt"].shape[0]
    assert n_local == 14850
    cu = torch.tensor([0, n_local + d["n_valid"], n_local + n_txt], dtype=torch.int32)
    out, _ = tr.infer(tw, img, d["txt"].cuda(), d["vec"].cuda(), cu, n_local + n_txt, (cos, sin))
    torch.cuda.synchronize()
    pa = tr.parallel_attention
    assert pa.copies == 0 and pa._buffers, "the fused driver must take the copy-free blocked exchange path"
    b = next(iter(pa._buffers.values()))
    assert tuple(b["snd"].shape) == (3, 8, 14850, 384), tuple(b["snd"].shape)
    assert out.shape[0] == n_local and torch.isfinite(out.float()).all()
    torch.save(out[d["local_rows"].cuda()].cpu(), os.path.join(out_dir, f"out_rank{r}.pt"))
    return "RANK_SHAPES_HUNYUAN_OK"


def main():
    dist.init_process_group("gloo")
    r, n = dist.get_rank(), dist.get_world_size()
    dist.all_to_all_single = _host_staged(dist.all_to_all_single)
    dist.all_gather_into_tensor = _host_staged(dist.all_gather_into_tensor)
    torch.cuda.set_device(0)
    from lightx2v_amd import lib

    lib.init(0)
    out_dir = os.environ["X2V_RANK_SHAPES_DIR"]
    d = torch.load(os.path.join(out_dir, "inputs.pt"), mmap=True)
    with torch.no_grad():
        tag = (wan if os.environ.get("X2V_RANK_SHAPES_MODEL", "wan") == "wan" else hunyuan)(r, n, d, out_dir)
    dist.barrier()
    if r == 0:
        print(tag)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
