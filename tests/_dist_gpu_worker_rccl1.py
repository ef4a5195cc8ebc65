"""world_size-1 worker for tests/test_gpu_dist.py: the process group is the real RCCL backend ("nccl" on ROCm) on the one GPU of
the box, so the collectives ulysses.py issues (all_to_all_single / all_gather_into_tensor on bf16 device tensors, on the
communication stream, joined by events) go through RCCL itself rather than the host-staged test shim of the world-2 workers.
With one rank every exchange is the identity, so the Ulysses-wrapped forward must equal the plain forward bit for bit.
The head->seq exchange is FORCED into its two-piece split-size form (`split_head2seq = "force"`: the self-exchange cut in two row
pieces), so that RCCL sees the `all_to_all_single(recv_view, send_view, out_split, in_split)` call signature of the N-GPU run, on the
communication stream, overlapped with the second piece's attention; the probe `split_form_ok` runs against RCCL as well.
Round 5 (VERDICT r4 next #7c): the WHOLE `attend_blocked` sequence is asserted, not only the split-size probe — per attention RCCL must have seen
3 seq->head exchanges of the blocked [1, S, hd] buffers and 2 head->seq pieces, all enqueued on the ONE communication stream (a spy records the
stream each collective was issued under), the copy-free path must have been taken (`pa.copies == 0`), bench.py's CommTimer must have bracketed
every exchange and every join with HIP events (counts and a positive total), and the CFG-branch two-stream driver (`cfg_branch_streams=True`: both
branches' exchanges feeding the same communication stream in host order) must give the same bits over RCCL — so that the only thing an N-GPU run
adds is N."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    from lightx2v_amd import lib, scheduler, synth, ulysses, wan

    lib.init(0)
    # the exchange primitives on their own
    g = torch.Generator().manual_seed(1)
    x = torch.randn(96, 4 * 128, generator=g).to(torch.bfloat16).cuda()
    assert torch.equal(ulysses.seq2head(x), x) and torch.equal(ulysses.head2seq(x), x)
    assert torch.equal(ulysses.post_process(ulysses.pre_process(x)), x)

    dims = dict(synth.WAN_DIMS["wan-tiny"], dim=512, num_heads=4, ffn_dim=1024, num_layers=2)
    ts = (16, 3, 12, 10)
    wd = synth.synth_wan_weights(dims, seed=3)
    lat, ctx, ctx_null = synth.synth_inputs(dims, ts)
    inputs = {"text_encoder_output": {"context": [c.cuda() for c in ctx], "context_null": [c.cuda() for c in ctx_null]}}
    assert ulysses.split_form_ok(None, torch.device("cuda", 0)), "RCCL rejected the split-size all_to_all_single form"
    outs = {}
    calls = []
    real_a2a = dist.all_to_all_single

    def spy(out, inp, out_split=None, in_split=None, **kw):
        calls.append((tuple(out.shape), tuple(inp.shape), out_split, in_split, torch.cuda.current_stream().cuda_stream))
        return real_a2a(out, inp, out_split, in_split, **kw)

    timers = {}
    for mode in ("single", "ulysses", "ulysses-split", "ulysses-split-cfg-streams"):
        cfg = wan.default_config(dims, target_shape=ts, target_video_length=9, infer_steps=4, parallel_attn_type=None if mode == "single" else "ulysses",
                                 cfg_branch_streams=(mode == "ulysses-split-cfg-streams"))
        model = wan.WanModel(cfg, {k: v.cuda() for k, v in wd.items()})
        if mode.startswith("ulysses-split"):
            pa = model.transformer_infer.parallel_attention
            pa.split_head2seq = "force"
            pa.comm_timer = timers[mode] = ulysses.CommTimer()
            calls.clear()
            dist.all_to_all_single = spy
        sch = scheduler.WanScheduler(cfg, device="cuda")
        sch.prepare(latents=lat)
        model.set_scheduler(sch)
        for i in range(2):
            sch.step_pre(i)
            model.infer(inputs)
            sch.step_post()
        outs[mode] = sch.latents.float().cpu()
        if mode != "single":
            pa = model.transformer_infer.parallel_attention
            assert pa.copies == 0 and pa._buffers, "the fused driver must take the copy-free blocked exchange path (attend_blocked)"
        if mode.startswith("ulysses-split"):
            dist.all_to_all_single = real_a2a
            n_attn = 2 * 2 * dims["num_layers"]  # forwards x steps x layers
            seq2head_calls = [c for c in calls if c[2] is None]
            split_calls = [c for c in calls if c[2] is not None]
            S_tok = synth.seq_len_of(ts)
            assert len(seq2head_calls) == 3 * n_attn and all(c[0] == (1, S_tok, dims["dim"]) == c[1] for c in seq2head_calls), (len(seq2head_calls), seq2head_calls[:2])
            assert len(split_calls) == 2 * n_attn and all(len(c[2]) == 1 and c[2] == c[3] and c[0][0] == c[2][0] for c in split_calls), split_calls[:4]
            comm_streams = {c[4] for c in calls}
            assert len(comm_streams) == 1 and comm_streams != {torch.cuda.default_stream().cuda_stream}, f"collectives issued under {len(comm_streams)} streams (one communication stream expected)"
            if mode == "ulysses-split-cfg-streams":
                il = model._cfg_interleave
                assert il._pa_b is not None and il._pa_b._buffers and il._pa_b.copies == 0 and il._pa_b.comm_stream is pa.comm_stream, "CFG-branch interleave: branch B must feed branch A's communication stream"
            torch.cuda.synchronize()
            t = timers[mode]
            t.enabled = False
            c_ms, e_ms, n_coll = t.totals_ms()
            # brackets: v's early exchange (1) + q/k (1, holding 2 collectives) + 2 head->seq pieces per attention, + 1 all_gather per forward
            assert n_coll == 4 * n_attn + 2 * 2 and c_ms > 0.0 and e_ms >= 0.0 and len(t.exposed) >= 2 * n_attn, (n_coll, c_ms, e_ms, len(t.exposed))
    dist.all_to_all_single = real_a2a
    assert torch.isfinite(outs["single"]).all()
    assert torch.equal(outs["single"], outs["ulysses"]), "world-1 Ulysses over RCCL differs from the plain forward"
    assert torch.equal(outs["single"], outs["ulysses-split"]), "world-1 Ulysses with the two-piece split-size head->seq exchange differs from the plain forward"
    assert torch.equal(outs["single"], outs["ulysses-split-cfg-streams"]), "world-1 Ulysses with the CFG branches on two compute streams differs from the plain forward"
    dist.barrier()
    torch.cuda.synchronize()
    print("DIST_GPU_RCCL1_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
