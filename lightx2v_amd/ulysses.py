"""Ulysses sequence parallelism for the Wan self-attention over RCCL/xGMI (one process per GPU,
`torch.distributed` backend "nccl" = RCCL on ROCm).

reference: lightx2v/attentions/distributed/ulysses/attn.py:7-91 (ulysses_attn), comm/all2all.py:6-89,
ulysses/wrap.py:53-71 (parallelize_wan), utils/wan/processor.py:9-37 (pre/post_process).

Same partitioning as the reference — tokens of the flattened (t,h,w) axis in N contiguous shards, weights
and the 512-token context replicated, only self-attention exchanges data — but laid out for the MI355X
node: xGMI is a full mesh of point-to-point links, so an all-to-all puts one message on every link at once
(7 x 12.1 MB at Wan-14B 720p, ≈79 us/link).  What the reference pays beyond the wire time is removed:
  * no host synchronisation (the reference calls torch.cuda.synchronize() twice per attention, attn.py:48,85);
    exchanges are stream-ordered on a side stream and joined with events;
  * NO layout copies between the kernels and the collectives (the fused driver's path, `attend_blocked`): the exchange buffers
    [N, S/N, (H/N)d] are kernel operands as they are.  v's projection writes the send buffer from its GEMM epilogue
    (x2v_gemm_bf16_blocked, N-blocked y), the q/k norm+RoPE kernel writes q's and k's (x2v_rmsnorm_rope_blocked_bf16), the received
    buffers ARE the attention's row-major [S, (H/N)d] operands, the attention output IS the head->seq send buffer, and the output
    projection reads the received [N, S/N, (H/N)d] buffer as a K-blocked x.  The reference transposes with a copy on both sides of
    both exchanges (all2all.py:29-33,41; :70-75,87);
  * v's exchange runs on the communication stream under the q and k projections and the norm+RoPE kernel; the head->seq exchange is
    issued in two halves (destination ranks [0, N/2) and [N/2, N)), the first under the attention of the second half's query rows.
`seq2head` / `head2seq` / calling the object with row-major 2-D q, k, v keep the reference's functional forms (bit-equal to its
all2all_* functions in the gloo tests) for callers that hold row-major tensors (the reference's op-by-op loop).
"""
import torch
import torch.distributed as dist

from . import lib
from .wan import CfgBranchStreams as _WanCfgBranchStreams


def _world(group=None):
    return dist.get_world_size(group), dist.get_rank(group)


def seq2head(x, group=None):
    """[S/N, H*d] (this rank's tokens, all heads) → [S, (H/N)*d] (all tokens, this rank's heads).
    Column block j of width H*d/N goes to rank j (reference: all2all.py:6-44)."""
    n, _ = _world(group)
    s_local, hd = x.shape
    send = x.view(s_local, n, hd // n).transpose(0, 1).contiguous()  # [N, S/N, hd/N]
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    return recv.view(n * s_local, hd // n)


def head2seq(x, group=None):
    """[S, (H/N)*d] → [S/N, H*d] (reference: all2all.py:47-89).  The send buffer is x itself."""
    n, _ = _world(group)
    s, hdn = x.shape
    recv = torch.empty((n, s // n, hdn), dtype=x.dtype, device=x.device)
    dist.all_to_all_single(recv, x.contiguous().view(n, s // n, hdn), group=group)
    return recv.transpose(0, 1).reshape(s // n, n * hdn)  # one gather-copy into token-major layout


def _split_plan(n, r, rows_per_rank, split):
    """head->seq pieces of an attention output o [N * rows_per_rank, hd/N] (row block j goes to rank j): a list of
    (destination ranks [j0, j1), send row range, in_split, out_split, recv row range of the [N * rows_per_rank] receive view).
    split False: one equal exchange.  True: two pieces by destination rank ([0, N/2) and [N/2, N)), the first flying under the attention
    of the second piece's query rows; a rank outside a piece's destinations receives nothing in it (zero-row receive view).
    "force" (tests, world size 1): the self-exchange cut in two row pieces, so that RCCL sees the split-size call signature of the
    N-GPU run on a one-GPU box."""
    S = n * rows_per_rank
    if n == 1 and split == "force":
        h = rows_per_rank // 2
        return [((0, 1), slice(0, h), [h], [h], slice(0, h)), ((0, 1), slice(h, S), [S - h], [S - h], slice(h, S))]
    if n < 2 or not split:
        return [((0, n), slice(0, S), None, None, slice(0, S))]
    plan = []
    for j0, j1 in ((0, n // 2), (n // 2, n)):
        mine = j0 <= r < j1
        plan.append(((j0, j1), slice(j0 * rows_per_rank, j1 * rows_per_rank), [rows_per_rank if j0 <= j < j1 else 0 for j in range(n)],
                     [rows_per_rank if mine else 0] * n, slice(0, S) if mine else slice(0, 0)))
    return plan


_SPLIT_PROBED = {}


def split_form_ok(group, device):
    """Does this PyTorch / RCCL build take the split-size `all_to_all_single` with a zero-row receive view (the two-piece head->seq
    exchange)?  Probed ONCE per (group, device) on a few rows, every rank running the same calls; a rank-local failure (argument checks
    raise before anything is sent) is agreed on with a MIN all-reduce so that all ranks take the same path afterwards.  X2V_ULYSSES_SPLIT=0/1
    skips the probe.  The fallback lives here, next to the collective, not only in bench.py."""
    import os

    env = os.environ.get("X2V_ULYSSES_SPLIT")
    if env is not None:
        return env != "0"
    key = (id(group), str(device))
    if key not in _SPLIT_PROBED:
        n, r = _world(group)
        ok = 1
        try:
            o = torch.zeros((n * 4, 8), dtype=torch.bfloat16, device=device)
            ro = torch.zeros_like(o)
            for _, rows, in_split, out_split, rrows in _split_plan(n, r, 4, True if n > 1 else "force"):
                dist.all_to_all_single(ro[rrows], o[rows], out_split, in_split, group=group)
            if o.is_cuda:
                torch.cuda.synchronize(device)
        except Exception as e:  # noqa: BLE001
            ok = 0
            print(f"lightx2v_amd.ulysses: split-size all_to_all_single rejected on rank {r} ({type(e).__name__}: {str(e)[:160]}); using one head->seq exchange", flush=True)
        flag = torch.tensor([ok], dtype=torch.int32, device=device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        _SPLIT_PROBED[key] = bool(flag.item())
    return _SPLIT_PROBED[key]


class CommTimer:
    """HIP-event accounting of the exchanges (bench.py, N > 1): every collective issued on the communication stream is bracketed by two events
    on that stream (`comm`: time the exchanges occupy the communication stream, queueing behind one another included), and every join of a
    compute stream behind the communication stream by two events on the compute stream (`exposed`: how long the compute stream had nothing to
    run but the wait — the part of the communication that no kernel of that stream hid; with the two CFG branches on two streams the other
    branch may still have been computing).  The reference pays for the same exchanges with two `torch.cuda.synchronize()` per attention
    (ulysses/attn.py:48,85); this is what replaced them costs.  Read after a device synchronisation."""

    def __init__(self):
        self.comm, self.exposed = [], []
        self.enabled = True

    def bracket(self, which, stream):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        (self.comm if which == "comm" else self.exposed).append((a, b))
        a.record(stream)
        return lambda: b.record(stream)

    def totals_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.comm), sum(a.elapsed_time(b) for a, b in self.exposed), len(self.comm)


class UlyssesAttention:
    """Callable injected as `transformer_infer.parallel_attention` (reference hook: transformer_infer.py:381-388)."""

    def __init__(self, group=None, attn_fn=None, overlap=True):
        self.group = group
        self._default_attn = attn_fn is None
        self.attn_fn = attn_fn or (lambda q, k, v, h, d, variant=0: lib.attention(q, k, v, h, d, variant=variant))
        self.overlap = overlap and torch.cuda.is_available()
        self.comm_stream = None
        self._buffers = {}
        self.copies = 0  # layout copies made by the row-major entry (the blocked entry makes none): asserted by the tests
        self.split_head2seq = self.split_head2seq_default  # head->seq in two halves (all_to_all_single with split sizes) under the second half's attention
        self.comm_timer = None  # bench.py: a CommTimer (shared by both CFG branches) while the timed region runs

    split_head2seq_default = True

    class _Pending:
        """An exchange already issued on the communication stream (result valid once the compute stream has joined it)."""

        def __init__(self, src, out):
            self.src, self.out = src, out

    # ---- the fused driver's path: exchange buffers are kernel operands ---------------------------------------------------
    def buffers(self, s_local, hd, dtype, device):
        """The six [N, S/N, hd/N] exchange buffers of one attention (send q/k/v, receive q/k/v), the attention output (= head->seq
        send buffer) and its receive buffer.  Allocated once per shape and reused by every layer: stream order makes that safe
        (a layer's exchanges have been joined by the compute stream before the next layer's kernels write the buffers)."""
        n, _ = _world(self.group)
        key = (s_local, hd, dtype, str(device))
        b = self._buffers.get(key)
        if b is None:
            mk = lambda: torch.empty((n, s_local, hd // n), dtype=dtype, device=device)  # noqa: E731
            b = self._buffers[key] = {name: mk() for name in ("sq", "sk", "sv", "rq", "rk", "rv", "o", "ro")}
        return b

    def _comm(self):
        if self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream()
        return self.comm_stream

    def _issue(self, cs, fn):
        """fn() = one or more collectives, enqueued on the communication stream `cs` (already ordered behind the compute stream by the caller)."""
        t = self.comm_timer
        with torch.cuda.stream(cs):
            done = t.bracket("comm", cs) if (t is not None and t.enabled) else None
            out = fn()
            if done is not None:
                done()
        return out

    def _join(self, cur, cs):
        """The compute stream `cur` waits for everything enqueued on the communication stream so far."""
        t = self.comm_timer
        done = t.bracket("exposed", cur) if (t is not None and t.enabled) else None
        cur.wait_stream(cs)
        if done is not None:
            done()

    def on_comm(self, fn, src):
        """Run the collective `fn()` (which reads `src` and returns a fresh tensor) on the communication stream, ordered behind what the current
        stream has enqueued, and join the current stream behind it.  EVERY collective of this driver goes through the one communication stream
        With synchronous collectives running on the calling stream, a gather issued from a compute stream would put kernels of the
        same RCCL communicator on several streams at once and leave their order to RCCL's internals."""
        if not (self.overlap and src.is_cuda):
            return fn()
        cur, cs = torch.cuda.current_stream(), self._comm()
        cs.wait_stream(cur)
        out = self._issue(cs, fn)
        self._join(cur, cs)
        src.record_stream(cs)   # produced under the compute stream, read by the collective
        out.record_stream(cur)  # allocated under the communication stream, read by the compute stream
        return out

    def begin_exchange_blocked(self, send, recv):
        """seq->head of one blocked send buffer [N, S/N, hd/N] into `recv` (same shape; viewed as row-major [S, hd/N] by the attention),
        on the communication stream when overlapping, stream-ordered after what the compute stream has enqueued so far."""
        if self.overlap and send.is_cuda:
            cs = self._comm()
            cs.wait_stream(torch.cuda.current_stream())
            self._issue(cs, lambda: dist.all_to_all_single(recv, send, group=self.group))
        else:
            dist.all_to_all_single(recv, send, group=self.group)
        return self._Pending(send, recv)

    def attend_blocked(self, bufs, num_heads, head_dim=128, timer=None, variant=0, v_pending=None):
        """q/k/v are in bufs["sq"/"sk"/"sv"] (v possibly already in flight: `v_pending`).  Returns bufs["ro"]: the received head->seq
        buffer [N, S/N, hd/N] = the output projection's K-blocked input."""
        n, r = _world(self.group)
        if num_heads % n != 0:
            raise lib.X2VError(f"Ulysses needs num_heads % world_size == 0 (H={num_heads}, N={n})")
        hl = num_heads // n
        s_local, hdn = bufs["sq"].shape[1], bufs["sq"].shape[2]
        S = n * s_local
        use_streams = self.overlap and bufs["sq"].is_cuda
        if use_streams:
            cur, cs = torch.cuda.current_stream(), self._comm()
            cs.wait_stream(cur)

            def qkv():
                dist.all_to_all_single(bufs["rq"], bufs["sq"], group=self.group)
                dist.all_to_all_single(bufs["rk"], bufs["sk"], group=self.group)
                if v_pending is None:
                    dist.all_to_all_single(bufs["rv"], bufs["sv"], group=self.group)

            self._issue(cs, qkv)
            self._join(cur, cs)
        else:
            dist.all_to_all_single(bufs["rq"], bufs["sq"], group=self.group)
            dist.all_to_all_single(bufs["rk"], bufs["sk"], group=self.group)
            if v_pending is None:
                dist.all_to_all_single(bufs["rv"], bufs["sv"], group=self.group)
        qh, kh, vh = (bufs[x].view(S, hdn) for x in ("rq", "rk", "rv"))
        o = bufs["o"].view(S, hdn)
        fast = self._default_attn and (variant & 0xFF) == lib.ATTN_FAST
        vt = lib.transpose_heads(vh, hl) if fast else None  # the ping-pong kernel reads V^T (1/N of the single-GPU transposition)
        # head->seq in two pieces by destination rank: rows of ranks [0, N/2) first, exchanged under the attention of the second piece
        split = self.split_head2seq
        if split is True and not split_form_ok(self.group, o.device):
            split = self.split_head2seq = False
        ro = bufs["ro"].view(S, hdn)
        for _, rows, in_split, out_split, rrows in _split_plan(n, r, s_local, split):

            def fn(rows=rows):
                if fast:
                    return lib.attention(qh[rows], kh, vh, hl, head_dim, variant=variant, vt=vt, out=o[rows])
                res = self.attn_fn(qh[rows], kh, vh, hl, head_dim, variant=variant) if variant else self.attn_fn(qh[rows], kh, vh, hl, head_dim)
                if res.data_ptr() != o[rows].data_ptr():
                    o[rows].copy_(res)
                return o[rows]

            timer("self", fn) if timer is not None else fn()
            if use_streams:
                cs.wait_stream(cur)
                self._issue(cs, lambda rows=rows, rrows=rrows, out_split=out_split, in_split=in_split: dist.all_to_all_single(ro[rrows], o[rows], out_split, in_split, group=self.group))
            else:
                dist.all_to_all_single(ro[rrows], o[rows], out_split, in_split, group=self.group)
        if use_streams:
            self._join(cur, cs)
        return bufs["ro"]

    # ---- row-major entry (the reference's functional form) -----------------------------------------------------------------
    def begin_exchange(self, x):
        """Start seq→head of a row-major `x` [S/N, H*d] on the communication stream now (stream-ordered after what has been enqueued
        on the compute stream so far); the caller keeps enqueueing independent work and hands the result to __call__."""
        if not (self.overlap and x.is_cuda):
            return None
        cs = self._comm()
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            out = seq2head(x, self.group)
        x.record_stream(cs)
        out.record_stream(cs)
        return self._Pending(x, out)

    def __call__(self, q, k, v, num_heads, head_dim=128, timer=None, variant=0):
        n, _ = _world(self.group)
        if num_heads % n != 0:
            raise lib.X2VError(f"Ulysses needs num_heads % world_size == 0 (H={num_heads}, N={n})")
        self.copies += 4  # three transposing sends + the gather of the received output
        if self.overlap and q.is_cuda:
            cs = self._comm()
            cur = torch.cuda.current_stream()
            cs.wait_stream(cur)
            with torch.cuda.stream(cs):
                qh, kh = seq2head(q, self.group), seq2head(k, self.group)
                vh = v.out if isinstance(v, self._Pending) else seq2head(v, self.group)
            for t in (q, k) + (() if isinstance(v, self._Pending) else (v,)):
                t.record_stream(cs)  # produced on the compute stream, read by the exchange
            cur.wait_stream(cs)
            for t in (qh, kh, vh):
                t.record_stream(cur)  # allocated under the communication stream, read by the attention kernel
        else:
            if isinstance(v, self._Pending):
                v = v.src
            qh, kh, vh = seq2head(q, self.group), seq2head(k, self.group), seq2head(v, self.group)
        if self._default_attn and (variant & 0xFF) == lib.ATTN_FAST:
            # the ping-pong kernel reads V^T: transposed before the timed launch, as on the single-GPU path (wan.py)
            vt = lib.transpose_heads(vh, num_heads // n)
            fn = lambda: lib.attention(qh, kh, vh, num_heads // n, head_dim, variant=variant, vt=vt)
        elif variant:
            fn = lambda: self.attn_fn(qh, kh, vh, num_heads // n, head_dim, variant=variant)
        else:
            fn = lambda: self.attn_fn(qh, kh, vh, num_heads // n, head_dim)
        o = timer("self", fn) if timer is not None else fn()
        return head2seq(o, self.group)


def pre_process(x, group=None):
    """Zero-pad S to a multiple of N and keep this rank's contiguous chunk (reference: processor.py:9-21)."""
    n, r = _world(group)
    pad = (n - x.shape[0] % n) % n
    if pad:
        x = torch.nn.functional.pad(x, (0, 0, 0, pad))
    return torch.chunk(x, n, dim=0)[r].contiguous()


def post_process(x, group=None):
    """all_gather of the block-stack output → [S_padded, D] (reference: processor.py:24-37)."""
    n, _ = _world(group)
    out = torch.empty((n * x.shape[0], x.shape[1]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out


def parallelize_wan(wan_model, group=None, attn_fn=None):
    """reference: ulysses/wrap.py:53-71 — swap the attention, shard x around the block stack.  Beyond the reference: with CFG the two
    forwards of a step CAN be driven block by block on two compute streams (config `cfg_branch_streams=True`) so that one branch's
    kernels run while the other waits for an exchange.  That form has only ever run on one GPU (gloo plumbing) — it is off by default ("auto" =
    one forward after the other under Ulysses) until a multi-GPU run has shown it safe and faster; bench.py times both forms at N > 1."""
    tr = wan_model.transformer_infer
    n, r = _world(group)
    tr.parallel_attention = UlyssesAttention(group, attn_fn)
    tr.sp_rank, tr.sp_world = r, n
    original_infer = tr.infer

    def new_infer(weights, grid_sizes, embed, x, embed0, seq_lens, freqs, context, audio_dit_blocks=None):
        x = pre_process(x, group)
        x = original_infer(weights, grid_sizes, embed, x, embed0, seq_lens, freqs, context)
        return tr.parallel_attention.on_comm(lambda: post_process(x, group), x)

    tr.infer = new_infer
    wan_model._cfg_interleave = CfgBranchStreams(wan_model, group, attn_fn)
    return wan_model


class CfgBranchStreams(_WanCfgBranchStreams):
    """The conditional and unconditional forwards of a CFG step under Ulysses, interleaved block by block on two compute streams
    (the stream / join mechanics are wan.CfgBranchStreams; this class adds the sharding and the per-branch exchange state).

    Sequentially (the reference: wan/model.py:197-226 runs one forward after the other, each block waiting for its all-to-alls with
    torch.cuda.synchronize(), ulysses/attn.py:48,85) every exchange a block cannot hide behind its own kernels is dead time on the GPU: the q / k
    seq->head exchange in front of the attention, the last head->seq piece behind it.  The two forwards share latents, timestep and weights
    and are independent until the CFG combine, so here block i of the unconditional branch is enqueued on a second stream right behind
    block i of the conditional one: whenever one branch's stream waits for its communication stream, the other's GEMMs / attention own
    the CUs — including the FFN-up GEMM of one branch under the head->seq exchange of the other.  Each branch has its own UlyssesAttention
    (exchange buffers); both feed ONE communication stream and one process group in host order (branch A's block i, then branch B's — the
    same on every rank), so RCCL is used exactly as in the sequential form.  Every kernel sees the same operands as in the sequential order:
    results are bit-identical (tests/_dist_gpu_worker.py)."""

    def __init__(self, wan_model, group=None, attn_fn=None):
        super().__init__(wan_model)
        self.group, self.attn_fn = group, attn_fn
        self._pa_a = self._pa_b = None

    def _attention_ok(self, tr):
        return hasattr(tr.parallel_attention, "attend_blocked")

    def _setup(self):
        if self._streams is None:
            pa_a = self.model.transformer_infer.parallel_attention
            # branch B: its own exchange buffers, but the SAME process group and the SAME communication stream as branch A — RCCL sees one
            # communicator fed from one stream in host order, exactly as in the sequential form (two communicators running concurrently on
            # one GPU would be a new way to deadlock that nothing here could test)
            self._pa_b = UlyssesAttention(self.group, self.attn_fn)
            self._pa_b.comm_stream = pa_a._comm()
        self._pa_a = self.model.transformer_infer.parallel_attention
        self._pa_b.split_head2seq = self._pa_a.split_head2seq
        self._pa_b.comm_timer = self._pa_a.comm_timer
        return super()._setup()

    def _shard(self, x):
        return pre_process(x, self.group)

    def _gather(self, x):
        # called under a branch's compute stream: the gather itself runs on the shared communication stream (UlyssesAttention.on_comm)
        return self._pa_a.on_comm(lambda: post_process(x, self.group), x)

    def _enter_branch(self, tr, b):
        tr.parallel_attention = self._pa_b if b else self._pa_a

    def _leave(self, tr):
        tr.parallel_attention = self._pa_a


# ------------------------------------------------------------------------------------------------ HunyuanVideo
def hunyuan_pre_process(latents, freqs_cos, freqs_sin, group=None):
    """reference: attentions/distributed/utils/hunyuan/processor.py:5-50 — split the latent grid along H (or W) and the
    RoPE tables the same way; text stays replicated.  Returns (latents, cos, sin, split_dim)."""
    n, r = _world(group)
    t, h, w = latents.shape[2], latents.shape[3] // 2, latents.shape[4] // 2
    if h % n == 0:
        split_dim = -2
    elif w % n == 0:
        split_dim = -1
    else:
        raise ValueError(f"Cannot split video sequence into world size ({n}) parts evenly")
    lat = torch.chunk(latents, n, dim=split_dim)[r].contiguous()
    d = freqs_cos.shape[-1]
    cos = torch.chunk(freqs_cos.reshape(t, h, w, d), n, dim=split_dim - 1)[r].reshape(-1, d).contiguous()
    sin = torch.chunk(freqs_sin.reshape(t, h, w, d), n, dim=split_dim - 1)[r].reshape(-1, d).contiguous()
    return lat, cos, sin, split_dim


def hunyuan_post_process(output, split_dim, group=None):
    """reference: processor.py:53-77 — all_gather the per-rank noise predictions and concatenate along the split axis."""
    n, _ = _world(group)
    out = torch.empty((n * output.shape[0], *output.shape[1:]), dtype=output.dtype, device=output.device)  # dim-0 concatenation
    dist.all_gather_into_tensor(out, output.contiguous(), group=group)
    return torch.cat(list(out.chunk(n, dim=0)), dim=split_dim)


class UlyssesHunyuanAttention:
    """reference: attentions/distributed/ulysses/attn.py:7-91 for the joint img+txt attention: image q/k/v go seq→head by
    all-to-all, the (replicated) text q/k/v contribute this rank's heads, one attention over [all image tokens ; text]
    with H/N heads, image output head→seq by all-to-all, text output all-gathered over heads.  No host synchronisation
    (the reference calls torch.cuda.synchronize() twice per attention, attn.py:48,85); receive buffers are row ranges of
    the joint attention operands, so there is no torch.cat."""

    def __init__(self, group=None, attn_fn=None, overlap=True):
        self.group = group
        self.attn_fn = attn_fn
        self.overlap = overlap and torch.cuda.is_available()
        self.comm_stream = None
        self._buffers = {}
        self.copies = 0  # image-row layout copies made by the row-major entry (the blocked entry makes none): asserted by the tests
        self.comm_timer = None  # tools/hunyuan_bench.py at N > 1: a CommTimer while the timed region runs (None: no events are recorded)
        self.split_head2seq = True

    # ---- the fused driver's path: exchange buffers are kernel operands ---------------------------------------------------
    def buffers(self, n_img, n_txt, hd, mlp, dtype, device):
        """Exchange buffers of one joint attention, allocated once per shape and reused by every block (stream order makes that safe):
          snd   [3, N, n_img, hd/N]      image q | k | v, head-blocked: written by the fused-QKV GEMM (N-blocked output, column block c of
                                         the 3*hd outputs = snd.view(3N, ...)[c]) and normalised / rotated in place
          joint [3, N*n_img + n_txt, hd/N] the attention operands: rows [0, N*n_img) receive the image exchange, the rest this rank's heads
                                         of the (replicated) text
          o     [N*n_img + n_txt, hd/N]   attention output; its image rows are the head->seq send buffer
          a_img [N + mlp/(hd/N), n_img, hd/N], a_txt [... , n_txt, ...]   K-blocked input of the projection that follows: blocks [0, N)
                                         receive head->seq (image) / the all-gather over heads (text); the further blocks hold the MLP branch
                                         of a single block (written N-blocked by linear1's GELU GEMM), so that linear2 reads
                                         cat(attn, mlp) (transformer_infer.py:377) without anybody assembling it."""
        n, _ = _world(self.group)
        key = (n_img, n_txt, hd, mlp, dtype, str(device))
        b = self._buffers.get(key)
        if b is None:
            hdn = hd // n
            e = lambda *sh: torch.empty(sh, dtype=dtype, device=device)  # noqa: E731
            b = self._buffers[key] = dict(snd=e(3, n, n_img, hdn), joint=e(3, n * n_img + n_txt, hdn), o=e(n * n_img + n_txt, hdn),
                                          a_img=e(n + mlp // hdn, n_img, hdn), a_txt=e(n + mlp // hdn, n_txt, hdn))
        return b

    def on_comm(self, fn, src):
        """As UlyssesAttention.on_comm: every collective of the driver on the one communication stream."""
        if not (self.overlap and src.is_cuda):
            return fn()
        if self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream()
        cur, cs = torch.cuda.current_stream(), self.comm_stream
        cs.wait_stream(cur)
        t = self.comm_timer if (self.comm_timer is not None and self.comm_timer.enabled) else None  # CommTimer accounting as in UlyssesAttention
        with torch.cuda.stream(cs):
            done = t.bracket("comm", cs) if t is not None else None
            out = fn()
            if done is not None:
                done()
        done = t.bracket("exposed", cur) if t is not None else None
        cur.wait_stream(cs)
        if done is not None:
            done()
        src.record_stream(cs)
        out.record_stream(cur)
        return out

    def blocked_ok(self, hd, mlp):
        """K-blocked GEMM operands advance in whole 64-element K tiles."""
        n, _ = _world(self.group)
        return hd % n == 0 and (hd // n) % 128 == 0 and mlp % (hd // n) == 0

    def attend_blocked(self, bufs, txt_qkv, segs_txt, num_heads, variant=0):
        """Image q/k/v are in bufs["snd"] (head-blocked), text q/k/v in the row-major views `txt_qkv` [n_txt, H*128] each.
        Returns (a_img, a_txt) whose first N blocks hold the attention output of this rank's image rows / of the text rows, all heads."""
        n, r = _world(self.group)
        if num_heads % n != 0:
            raise lib.X2VError(f"Ulysses needs num_heads % world_size == 0 (H={num_heads}, N={n})")
        snd, joint, o, a_img, a_txt = (bufs[x] for x in ("snd", "joint", "o", "a_img", "a_txt"))
        n_img, hdn = snd.shape[2], snd.shape[3]
        n_txt, tot = a_txt.shape[1], n * snd.shape[2]
        n_valid = segs_txt[0]
        hl = num_heads // n
        use_streams = self.overlap and snd.is_cuda
        cur = cs = None
        if use_streams:
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream()
            cur, cs = torch.cuda.current_stream(), self.comm_stream

        tm = self.comm_timer if (use_streams and self.comm_timer is not None and self.comm_timer.enabled) else None

        def on_comm(fn):
            if use_streams:
                cs.wait_stream(cur)
                with torch.cuda.stream(cs):
                    done = tm.bracket("comm", cs) if tm is not None else None
                    fn()
                    if done is not None:
                        done()
            else:
                fn()

        def join():  # the compute stream waits for everything enqueued on the communication stream so far
            done = tm.bracket("exposed", cur) if tm is not None else None
            cur.wait_stream(cs)
            if done is not None:
                done()

        def seq2head():
            for i in range(3):
                dist.all_to_all_single(joint[i][:tot].view(n, n_img, hdn), snd[i], group=self.group)

        on_comm(seq2head)
        for i in range(3):  # this rank's heads of the text tokens (n_txt x hd/N: a few hundred rows), beside the exchange
            joint[i][tot:].copy_(txt_qkv[i][:, r * hdn : (r + 1) * hdn])
        if use_streams:
            join()
        jq, jk, jv = joint[0], joint[1], joint[2]
        nq = tot + n_valid
        fast = self.attn_fn is None and (variant & 0xFF) == lib.ATTN_FAST
        vt = lib.transpose_heads(jv[:nq], hl) if fast else None

        def attend(rows):
            if fast:
                lib.attention(jq[rows], jk[:nq], jv[:nq], hl, 128, out=o[rows], variant=variant, vt=vt)
            elif self.attn_fn is not None:
                self.attn_fn(jq[rows], jk[:nq], jv[:nq], hl, o[rows])
            else:
                lib.attention(jq[rows], jk[:nq], jv[:nq], hl, 128, out=o[rows], variant=variant)

        # head->seq of the image rows in two pieces by destination rank: the first piece's exchange runs under the second piece's attention
        split = self.split_head2seq
        if split is True and not split_form_ok(self.group, o.device):
            split = self.split_head2seq = False
        plan = _split_plan(n, r, n_img, split)
        recv_all = a_img[:n].view(tot, hdn)
        for pi, (_, rows, in_split, out_split, rrows) in enumerate(plan):
            last = pi == len(plan) - 1
            attend(slice(rows.start, nq if last else rows.stop))  # the valid text queries ride with the last piece
            if last and n_valid < n_txt:  # the padded text tokens attend among themselves (second cu_seqlens segment)
                pad = slice(nq, tot + n_txt)
                if self.attn_fn is not None:
                    self.attn_fn(jq[pad], jk[pad], jv[pad], hl, o[pad])
                else:
                    lib.attention(jq[pad], jk[pad], jv[pad], hl, 128, out=o[pad], variant=variant)
            on_comm(lambda recv=recv_all[rrows], send=o[rows], out_split=out_split, in_split=in_split: dist.all_to_all_single(recv, send, out_split, in_split, group=self.group))
        # text rows: every rank holds all text tokens for its heads -> gather the head blocks (block j = rank j's heads)
        on_comm(lambda: dist.all_gather_into_tensor(a_txt[:n].view(n * n_txt, hdn), o[tot:], group=self.group))
        if use_streams:
            join()
        return a_img, a_txt

    # ---- row-major entry (the reference's functional form) -----------------------------------------------------------------
    def __call__(self, q, k, v, n_img, segs_txt, num_heads, out, variant=0):
        """q, k, v: [n_img_local + n_txt, H*128] views; out: same rows, H*128 columns.  segs_txt = (n_valid_txt, n_txt)."""
        n, r = _world(self.group)
        if num_heads % n != 0:
            raise lib.X2VError(f"Ulysses needs num_heads % world_size == 0 (H={num_heads}, N={n})")
        self.copies += 4  # three transposing sends + the gather of the received output
        hd = q.shape[1]
        hdn = hd // n
        n_txt = q.shape[0] - n_img
        tot = n * n_img
        joint = [torch.empty((tot + n_txt, hdn), dtype=q.dtype, device=q.device) for _ in range(3)]
        for src, dst in zip((q, k, v), joint):
            send = src[:n_img].reshape(n_img, n, hdn).transpose(0, 1).contiguous()  # [N, n_img, hd/N]
            dist.all_to_all_single(dst[:tot].view(n, n_img, hdn), send, group=self.group)
            dst[tot:].copy_(src[n_img:, r * hdn : (r + 1) * hdn])
        o = torch.empty((tot + n_txt, hdn), dtype=q.dtype, device=q.device)
        n_valid = segs_txt[0]
        fn = self.attn_fn or (lambda a, b, c, h, oo: lib.attention(a, b, c, h, 128, out=oo, variant=variant))
        fn(joint[0][: tot + n_valid], joint[1][: tot + n_valid], joint[2][: tot + n_valid], num_heads // n, o[: tot + n_valid])
        if n_valid < n_txt:
            fn(joint[0][tot + n_valid :], joint[1][tot + n_valid :], joint[2][tot + n_valid :], num_heads // n, o[tot + n_valid :])
        # image rows: head→seq; the send buffer [N, n_img, hd/N] is o's image part as it is
        recv = torch.empty((n, n_img, hdn), dtype=q.dtype, device=q.device)
        dist.all_to_all_single(recv, o[:tot].view(n, n_img, hdn), group=self.group)
        out[:n_img].view(n_img, n, hdn).copy_(recv.transpose(0, 1))
        # text rows: every rank holds all text tokens for its heads → gather the head blocks
        gathered = torch.empty((n * n_txt, hdn), dtype=q.dtype, device=q.device)
        dist.all_gather_into_tensor(gathered, o[tot:].contiguous(), group=self.group)
        out[n_img:].view(n_txt, n, hdn).copy_(gathered.view(n, n_txt, hdn).transpose(0, 1))
        return out


def parallelize_hunyuan(hunyuan_model, group=None, attn_fn=None):
    """reference: ulysses/wrap.py:5-50 — shard latents + RoPE tables around HunyuanModel.infer, swap the attention."""
    hunyuan_model.transformer_infer.parallel_attention = UlyssesHunyuanAttention(group, attn_fn)
    original_infer = hunyuan_model.infer

    def new_infer(inputs):
        sch = hunyuan_model.scheduler
        keep = (sch.latents, sch.freqs_cos, sch.freqs_sin)
        sch.latents, sch.freqs_cos, sch.freqs_sin, split_dim = hunyuan_pre_process(*keep, group=group)
        original_infer(inputs)
        pa = hunyuan_model.transformer_infer.parallel_attention
        sch.noise_pred = pa.on_comm(lambda: hunyuan_post_process(sch.noise_pred, split_dim, group), sch.noise_pred)
        sch.latents, sch.freqs_cos, sch.freqs_sin = keep

    hunyuan_model.infer = new_infer
    return hunyuan_model
