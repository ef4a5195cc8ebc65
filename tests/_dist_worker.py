"""world_size-2 gloo worker for tests/test_dist_cpu.py (CPU): checks the Ulysses exchange logic of
lightx2v_amd.ulysses against single-process results, with the CPU oracle as the attention function."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    dist.init_process_group("gloo")
    r, n = dist.get_rank(), dist.get_world_size()
    from lightx2v_amd import ulysses
    from oracle import wan_oracle as O

    gen = torch.Generator().manual_seed(0)
    S, H, d = 96, 4, 128
    q, k, v = (torch.randn(S, H * d, generator=gen).to(torch.bfloat16) for _ in range(3))
    full = O.sdpa(q.view(S, H, d), k.view(S, H, d), v.view(S, H, d))
    sl = slice(r * S // n, (r + 1) * S // n)

    # exchange primitives are exact permutations
    qh = ulysses.seq2head(q[sl].contiguous())
    assert torch.equal(qh, q[:, r * H * d // n : (r + 1) * H * d // n])
    back = ulysses.head2seq(qh)
    assert torch.equal(back, q[sl])

    attn = ulysses.UlyssesAttention(attn_fn=lambda a, b, c, h, hd: O.sdpa(a.view(-1, h, hd), b.view(-1, h, hd), c.view(-1, h, hd)), overlap=False)
    out = attn(q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous(), H, d)
    assert torch.equal(out, full[sl]), (out.float() - full[sl].float()).abs().max()

    # shard / gather around the block stack, including zero padding when S % N != 0
    x = torch.randn(S + 1, 8, generator=gen)
    xs = ulysses.pre_process(x)
    assert xs.shape[0] == (S + 2) // n
    g = ulysses.post_process(xs)
    assert torch.equal(g[: S + 1], x) and torch.equal(g[S + 1 :], torch.zeros(1, 8))

    # against the reference's own functions in this real 2-process run (authoring container only: /root/reference is absent elsewhere):
    # comm/all2all.py:6-89 (seq<->head all-to-all layouts) and utils/wan/processor.py:9-37 (shard / gather)
    from oracle import ref_import

    if ref_import.reference_available():
        ref_import.patch_and_import()
        from lightx2v.attentions.distributed.comm.all2all import all2all_head2seq, all2all_seq2head
        from lightx2v.attentions.distributed.utils.wan import processor as ref_proc

        mine = ulysses.seq2head(q[sl].contiguous())
        theirs = all2all_seq2head(q[sl].contiguous().view(-1, H, d))
        assert torch.equal(mine.view(S, H // n, d), theirs), "seq2head differs from the reference's all2all_seq2head"
        o_h = torch.randn(S, (H // n) * d, generator=torch.Generator().manual_seed(5 + r)).to(torch.bfloat16)
        assert torch.equal(ulysses.head2seq(o_h).view(-1, H, d), all2all_head2seq(o_h.view(S, H // n, d))), "head2seq differs from all2all_head2seq"
        xr = torch.randn(S + 1, 8, generator=torch.Generator().manual_seed(3))
        a, b = ulysses.pre_process(xr), ref_proc.pre_process(xr)
        assert torch.equal(a, b)
        assert torch.equal(ulysses.post_process(a), ref_proc.post_process(b.contiguous()))
        if r == 0:
            print("REFERENCE_EXCHANGE_OK")

    # distributed oracle forward == single-process oracle forward (pins compute_freqs_dist + sharding; S % N == 0)
    from lightx2v_amd import synth

    dims = synth.WAN_DIMS["wan-tiny"]
    wd = synth.synth_wan_weights(dims, seed=0)
    lat, ctx, _ = synth.synth_inputs(dims, (16, 3, 8, 8))
    t = torch.tensor(500)
    embed, grid, xfull, embed0, s, context = O.wan_pre_infer(wd, dims, lat.to(torch.bfloat16), t, ctx)
    freqs = O.rope_freqs_table(128)
    ref = xfull.clone()
    for i in range(dims["num_layers"]):
        ref = O.wan_block(wd, i, dims, grid, ref, embed0, freqs, context)
    xs = ulysses.pre_process(xfull)
    sp_attn = lambda a, b, c: attn(a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1), c.reshape(c.shape[0], -1), dims["num_heads"], 128)
    for i in range(dims["num_layers"]):
        xs = O.wan_block(wd, i, dims, grid, xs, embed0, freqs, context, sp=(r, n, sp_attn))
    got = ulysses.post_process(xs)
    # sharded GEMMs run with a different M, so oneDNN may block/sum differently: agreement to bf16 rounding, not bits
    rel = ((got.float() - ref.float()).norm() / ref.float().norm()).item()
    assert rel < 1e-2 and (got.float() - ref.float()).abs().max() <= 2 ** -3, rel
    dist.barrier()
    if r == 0:
        print("DIST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
