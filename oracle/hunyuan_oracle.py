"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path `lightx2v_amd/`).

CPU restatement, in plain PyTorch, of the reference's HunyuanVideo DiT forward (DTYPE=BF16, t2v) — the algorithm
`lightx2v_amd/hunyuan.py` must reproduce.  Paths below are relative to /root/reference/lightx2v/.

  pre-infer     models/networks/hunyuan/infer/pre_infer.py:6-154
  double block  models/networks/hunyuan/infer/transformer_infer.py:81-310
  single block  models/networks/hunyuan/infer/transformer_infer.py:312-384
  RoPE          models/networks/hunyuan/infer/utils_bf16.py:5-31 (real cos/sin form, bf16 arithmetic)
  post-infer    models/networks/hunyuan/infer/post_infer.py:4-33
  scheduler     models/schedulers/hunyuan/scheduler.py:18-63,66-108,111-172,175-179,237-260,278-319

Pinning (tests/test_oracle_golden.py): `tests/golden/hunyuan_tiny.safetensors` is generated FROM THE REFERENCE's own
HunyuanPreInfer / HunyuanTransformerInfer / HunyuanPostInfer objects (oracle/gen_golden.py::gen_hunyuan) at a reduced
width; this file must reproduce it bit-exactly.  Two reference calls cannot be executed as shipped and are pinned
through a documented substitute in the generator: (1) `txt_in_attn_1` (pre_infer.py:117-119,140) passes 4-D q/k/v to
TorchSDPAWeight.apply, which only accepts 3-D at this snapshot (attn_weight.py:229-239) — the generator swaps in the
4-D form of the same SDPA call; (2) the scheduler module imports `diffusers` (absent) for `randn_tensor` only — a stub
provides that symbol, everything checked here (sigmas, timesteps, RoPE tables, Euler step) is the reference's code.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle.wan_oracle import _ACT, _sdpa_exact_chunked, layer_norm, mm, rms_norm

BF16 = torch.bfloat16


def _act():
    """Activation dtype of the graph: bf16, or what `wan_oracle.truth_precision()` switched to (the same statements evaluated without the bf16
    rounding points — the accuracy reference of the full-size model test).  Every tensor this module creates lives on its inputs' device, so the
    statements also run through plain PyTorch on a GPU (tests/test_gpu_full_size.py: 119 056 tokens x 60 blocks are out of the host's reach)."""
    return _ACT[-1]


# ----------------------------------------------------------------------------- scheduler pieces
def set_timesteps_sigmas(num_inference_steps, shift, num_train_timesteps=1000):
    """schedulers/hunyuan/scheduler.py:175-179."""
    sigmas = torch.linspace(1, 0, num_inference_steps + 1)
    sigmas = (shift * sigmas) / (1 + (shift - 1) * sigmas)
    return (sigmas[:-1] * num_train_timesteps).to(torch.float32), sigmas


def rope_tables(rope_sizes, rope_dim_list=(16, 56, 56), theta=256.0, dtype=BF16):
    """get_nd_rotary_pos_embed(use_real=True) (scheduler.py:18-63,66-108,111-172) as called by
    prepare_rotary_pos_embedding (:278-319): per axis a, freqs = 1/theta^(2j/d_a), cos/sin of pos*freqs with every value
    repeated twice (interleaved pairs), axes concatenated → [T*H*W, 128] each, then rounded to bf16 (`dtype`: the truth evaluation keeps fp32)."""
    axes = [torch.linspace(0, n, n + 1, dtype=torch.float32)[:n] for n in rope_sizes]
    grid = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=0)
    cos, sin = [], []
    for a, d in enumerate(rope_dim_list):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2)[: d // 2].float() / d))
        f = torch.outer(grid[a].reshape(-1), freqs)
        cos.append(f.cos().repeat_interleave(2, dim=1))
        sin.append(f.sin().repeat_interleave(2, dim=1))
    return torch.cat(cos, dim=1).to(dtype), torch.cat(sin, dim=1).to(dtype)


def euler_step(latents, noise_pred, sigmas, step_index):
    """HunyuanScheduler.step_post, t2v (scheduler.py:256-260)."""
    dt = sigmas[step_index + 1] - sigmas[step_index]
    return latents.to(torch.float32) + noise_pred.to(torch.float32) * dt


# ----------------------------------------------------------------------------- RoPE on q, k
def apply_rotary_emb(xq, xk, cos, sin):
    """utils_bf16.py:11-31: x*cos + rotate_half(x)*sin with rotate_half(x)[2i] = -x[2i+1], [2i+1] = x[2i]; bf16 tensors,
    so the two products and the sum each round to bf16.  xq/xk [L, H, D]; cos/sin [L, D]."""
    L, H, D = xq.shape
    c, s = cos.view(L, 1, D), sin.view(L, 1, D)

    def rot(x):
        re, im = x.reshape(L, H, -1, 2).unbind(-1)
        return torch.stack([-im, re], dim=-1).flatten(2)

    return xq * c + rot(xq) * s, xk * c + rot(xk) * s


def varlen_attention(q, k, v, cu_seqlens):
    """What `flash_attn_varlen_func` computes for weights.double_attn / single_attn (attn_weight.py:76-126 with the
    cu_seqlens built in pre_infer.py:50-56): dense non-causal attention inside each [cu[i], cu[i+1]) segment.
    q, k, v [L, H, D] → [L, H*D].  With an all-ones text mask the second segment is empty and this equals the
    `torch_sdpa` op's dense attention (the form the fixture was generated with)."""
    out = torch.empty(q.shape[0], q.shape[1] * q.shape[2], dtype=q.dtype, device=q.device)
    for a, b in zip(cu_seqlens[:-1].tolist(), cu_seqlens[1:].tolist()):
        if b > a:
            if q.dtype != BF16 and (b - a) * (b - a) * q.shape[1] * q.element_size() > (32 << 30):
                out[a:b] = _sdpa_exact_chunked(q[a:b], k[a:b], v[a:b])  # truth evaluation at a size whose score tensor does not fit in one piece
                continue
            qs, ks, vs = (t[a:b].unsqueeze(0).transpose(1, 2) for t in (q, k, v))
            o = F.scaled_dot_product_attention(qs, ks, vs).transpose(1, 2)
            out[a:b] = o.reshape(b - a, -1)
    return out


def _split_heads(qkv, heads):
    """rearrange(qkv, "L (K H D) -> K L H D", K=3, H=heads)."""
    L = qkv.shape[0]
    return qkv.view(L, 3, heads, -1).permute(1, 0, 2, 3)


def _lin(wd, name, x):
    return mm(x, wd[name + ".weight"], wd[name + ".bias"])


# ----------------------------------------------------------------------------- blocks
def double_block(wd, i, img, txt, vec, freqs, heads, cu_seqlens):
    """transformer_infer.py:81-310 (t2v: token_replace_vec is None)."""
    p = f"double_blocks.{i}."
    vec_silu = F.silu(vec)
    i_sh1, i_sc1, i_g1, i_sh2, i_sc2, i_g2 = _lin(wd, p + "img_mod.linear", vec_silu).chunk(6, dim=-1)
    t_sh1, t_sc1, t_g1, t_sh2, t_sc2, t_g2 = _lin(wd, p + "txt_mod.linear", vec_silu).chunk(6, dim=-1)

    x = layer_norm(img) * (1 + i_sc1) + i_sh1
    iq, ik, iv = _split_heads(_lin(wd, p + "img_attn_qkv", x), heads)
    iq, ik = rms_norm(iq, wd[p + "img_attn_q_norm.weight"]), rms_norm(ik, wd[p + "img_attn_k_norm.weight"])
    iq, ik = apply_rotary_emb(iq, ik, *freqs)
    y = layer_norm(txt) * (1 + t_sc1) + t_sh1
    tq, tk, tv = _split_heads(_lin(wd, p + "txt_attn_qkv", y), heads)
    tq, tk = rms_norm(tq, wd[p + "txt_attn_q_norm.weight"]), rms_norm(tk, wd[p + "txt_attn_k_norm.weight"])

    attn = varlen_attention(torch.cat((iq, tq), 0), torch.cat((ik, tk), 0), torch.cat((iv, tv), 0), cu_seqlens)
    n_img = img.shape[0]
    img_out = _lin(wd, p + "img_attn_proj", attn[:n_img])
    txt_out = _lin(wd, p + "txt_attn_proj", attn[n_img:])

    img = img + img_out * i_g1
    x = layer_norm(img) * (1 + i_sc2) + i_sh2
    x = _lin(wd, p + "img_mlp.fc2", F.gelu(_lin(wd, p + "img_mlp.fc1", x), approximate="tanh"))
    txt = txt + txt_out * t_g1
    y = layer_norm(txt) * (1 + t_sc2) + t_sh2
    y = _lin(wd, p + "txt_mlp.fc2", F.gelu(_lin(wd, p + "txt_mlp.fc1", y), approximate="tanh"))
    return img + x * i_g2, txt + y * t_g2


def single_block(wd, i, x, vec, txt_len, freqs, heads, hidden, cu_seqlens):
    """transformer_infer.py:312-384."""
    p = f"single_blocks.{i}."
    shift, scale, gate = _lin(wd, p + "modulation.linear", F.silu(vec)).chunk(3, dim=-1)
    xm = _lin(wd, p + "linear1", layer_norm(x) * (1 + scale) + shift)
    qkv, mlp = xm[:, : 3 * hidden], xm[:, 3 * hidden :]
    q, k, v = _split_heads(qkv, heads)
    q, k = rms_norm(q, wd[p + "q_norm.weight"]), rms_norm(k, wd[p + "k_norm.weight"])
    iq, ik = apply_rotary_emb(q[:-txt_len], k[:-txt_len], *freqs)
    q, k = torch.cat((iq, q[-txt_len:]), 0), torch.cat((ik, k[-txt_len:]), 0)
    attn = varlen_attention(q, k, v, cu_seqlens)
    out = _lin(wd, p + "linear2", torch.cat((attn, F.gelu(mlp, approximate="tanh")), 1))
    return x + out * gate


def transformer_infer(wd, dims, img, txt, vec, cu_seqlens, freqs):
    """transformer_infer.py:66-79 (_infer_without_offload)."""
    for i in range(dims["double_blocks"]):
        img, txt = double_block(wd, i, img, txt, vec, freqs, dims["heads"], cu_seqlens)
    x = torch.cat((img, txt), 0)
    for i in range(dims["single_blocks"]):
        x = single_block(wd, i, x, vec, txt.shape[0], freqs, dims["heads"], dims["hidden"], cu_seqlens)
    return x[: img.shape[0]], vec


# ----------------------------------------------------------------------------- blocks on a row subset (full-size parity)
def _rope1(x, cos, sin):
    """apply_rotary_emb for one tensor (same statements; utils_bf16.py:11-31)."""
    L, H, D = x.shape
    re, im = x.reshape(L, H, -1, 2).unbind(-1)
    return x * cos.view(L, 1, D) + torch.stack([-im, re], dim=-1).flatten(2) * sin.view(L, 1, D)


def _attend_rows(q_sel, sel_pos, k, v, cu_seqlens):
    """varlen attention for selected query rows: q_sel [n, H, D] sit at joint positions `sel_pos` (LongTensor); k / v are ALL joint rows.
    A query attends the keys of its own [cu[i], cu[i+1]) segment."""
    out = torch.empty(q_sel.shape[0], q_sel.shape[1] * q_sel.shape[2], dtype=q_sel.dtype)
    for a, b in zip(cu_seqlens[:-1].tolist(), cu_seqlens[1:].tolist()):
        m = (sel_pos >= a) & (sel_pos < b)
        if b > a and bool(m.any()):
            o = F.scaled_dot_product_attention(q_sel[m].unsqueeze(0).transpose(1, 2), k[a:b].unsqueeze(0).transpose(1, 2), v[a:b].unsqueeze(0).transpose(1, 2)).transpose(1, 2)
            out[m] = o.reshape(int(m.sum()), -1)
    return out


def double_block_rows(wd, i, img, txt, vec, freqs, heads, cu_seqlens, rows):
    """`double_block` for a SUBSET of the image rows (all text rows): every op is row-wise except the joint attention, whose keys / values need
    all rows — LayerNorm + modulate and the k / v thirds of `img_attn_qkv` run on all image rows, everything else on `rows` (LongTensor of
    image-row indices).  Returns (img_out[rows], txt_out).  Same statements as `double_block` (transformer_infer.py:81-310); the CPU suite
    requires it to equal double_block(...)[0][rows] to rounding."""
    p = f"double_blocks.{i}."
    D = img.shape[1]
    vec_silu = F.silu(vec)
    i_sh1, i_sc1, i_g1, i_sh2, i_sc2, i_g2 = _lin(wd, p + "img_mod.linear", vec_silu).chunk(6, dim=-1)
    t_sh1, t_sc1, t_g1, t_sh2, t_sc2, t_g2 = _lin(wd, p + "txt_mod.linear", vec_silu).chunk(6, dim=-1)
    cos, sin = freqs
    x = layer_norm(img) * (1 + i_sc1) + i_sh1
    W, B = wd[p + "img_attn_qkv.weight"], wd[p + "img_attn_qkv.bias"]
    kv = mm(x, W[D:], B[D:])
    L = img.shape[0]
    ik = _rope1(rms_norm(kv[:, :D].reshape(L, heads, -1), wd[p + "img_attn_k_norm.weight"]), cos, sin)
    iv = kv[:, D:].reshape(L, heads, -1)
    iq = _rope1(rms_norm(mm(x[rows], W[:D], B[:D]).reshape(len(rows), heads, -1), wd[p + "img_attn_q_norm.weight"]), cos[rows], sin[rows])
    y = layer_norm(txt) * (1 + t_sc1) + t_sh1
    tq, tk, tv = _split_heads(_lin(wd, p + "txt_attn_qkv", y), heads)
    tq, tk = rms_norm(tq, wd[p + "txt_attn_q_norm.weight"]), rms_norm(tk, wd[p + "txt_attn_k_norm.weight"])
    n_txt = txt.shape[0]
    pos = torch.cat((rows, L + torch.arange(n_txt)))
    attn = _attend_rows(torch.cat((iq, tq), 0), pos, torch.cat((ik, tk), 0), torch.cat((iv, tv), 0), cu_seqlens)
    n = len(rows)
    img_r = img[rows] + _lin(wd, p + "img_attn_proj", attn[:n]) * i_g1
    xr = layer_norm(img_r) * (1 + i_sc2) + i_sh2
    xr = _lin(wd, p + "img_mlp.fc2", F.gelu(_lin(wd, p + "img_mlp.fc1", xr), approximate="tanh"))
    txt_n = txt + _lin(wd, p + "txt_attn_proj", attn[n:]) * t_g1
    yy = layer_norm(txt_n) * (1 + t_sc2) + t_sh2
    yy = _lin(wd, p + "txt_mlp.fc2", F.gelu(_lin(wd, p + "txt_mlp.fc1", yy), approximate="tanh"))
    return img_r + xr * i_g2, txt_n + yy * t_g2


def single_block_rows(wd, i, x, vec, txt_len, freqs, heads, hidden, cu_seqlens, rows):
    """`single_block` for a subset of the IMAGE rows plus all text rows (transformer_infer.py:312-384): returns the block's output at joint
    positions cat(rows, text rows).  The k / v thirds of linear1 run on all rows."""
    p = f"single_blocks.{i}."
    shift, scale, gate = _lin(wd, p + "modulation.linear", F.silu(vec)).chunk(3, dim=-1)
    L = x.shape[0]
    n_img = L - txt_len
    sel = torch.cat((rows, n_img + torch.arange(txt_len)))
    xn = layer_norm(x) * (1 + scale) + shift
    W, B = wd[p + "linear1.weight"], wd[p + "linear1.bias"]
    kv = mm(xn, W[hidden : 3 * hidden], B[hidden : 3 * hidden])
    k = rms_norm(kv[:, :hidden].reshape(L, heads, -1), wd[p + "k_norm.weight"])
    v = kv[:, hidden:].reshape(L, heads, -1)
    cos, sin = freqs
    k = torch.cat((_rope1(k[:n_img], cos, sin), k[n_img:]), 0)
    xs = xn[sel]
    q = rms_norm(mm(xs, W[:hidden], B[:hidden]).reshape(len(sel), heads, -1), wd[p + "q_norm.weight"])
    q = torch.cat((_rope1(q[: len(rows)], cos[rows], sin[rows]), q[len(rows) :]), 0)
    mlp = mm(xs, W[3 * hidden :], B[3 * hidden :])
    attn = _attend_rows(q, sel, k, v, cu_seqlens)
    out = _lin(wd, p + "linear2", torch.cat((attn, F.gelu(mlp, approximate="tanh")), 1))
    return x[sel] + out * gate


class TeaCacheOracle:
    """hunyuan/infer/feature_caching/transformer_infer.py:7-135 (HunyuanTransformerInferTeaCaching) restated.  After every forward the
    first double block's modulated input of the RETURNED image tokens decides whether the NEXT step runs the block stack: the polynomial-
    rescaled relative L1 change against the previous step's modulated input is accumulated; below `thresh` the next step re-applies the cached
    residual img_out - img_in instead.  The first and the last step always compute.  Quirk kept (:21): the modulation is `img_mod(vec)`
    WITHOUT the SiLU the blocks apply in front of it.  Pinned bit-exactly to tests/golden/hunyuan_teacache.safetensors."""

    COEFFICIENTS = [7.33226126e02, -4.01131952e02, 6.75869174e01, -3.14987800e00, 9.61237896e-02]

    def __init__(self, infer_steps, thresh):
        self.infer_steps, self.thresh = infer_steps, thresh
        self.records = [True] * infer_steps
        self.accumulated = 0
        self.prev_mod = None
        self.prev_res = None

    def should_calc(self, wd, step_index, img, vec):  # calculate_should_calc :17-43
        sh, sc, _, _, _, _ = _lin(wd, "double_blocks.0.img_mod.linear", vec.clone()).chunk(6, dim=-1)
        mod = F.layer_norm(img.clone(), (img.shape[1],), None, None, 1e-6) * (1 + sc) + sh
        if step_index == 0 or step_index == self.infer_steps - 1:
            calc = True
            self.accumulated = 0
        else:
            self.accumulated += np.poly1d(self.COEFFICIENTS)(((mod - self.prev_mod).abs().mean() / self.prev_mod.abs().mean()).cpu().item())
            calc = not (self.accumulated < self.thresh)
            if calc:
                self.accumulated = 0
        self.prev_mod = mod
        return calc

    def infer(self, wd, dims, step_index, img, txt, vec, cu_seqlens, freqs):  # infer :45-58
        if self.records[step_index]:
            ori = img.clone()
            img, vec = transformer_infer(wd, dims, img, txt, vec, cu_seqlens, freqs)
            self.prev_res = img - ori
        else:
            img = img + self.prev_res  # `img += previous_residual` (:121)
        if step_index <= self.infer_steps - 2:
            self.records[step_index + 1] = self.should_calc(wd, step_index, img, vec)
        return img, vec


# ----------------------------------------------------------------------------- pre / post
def _t_embed(t, device=None):
    """pre_infer.py:62-64,73-75,147-149: cos|sin of t * exp(-ln(1e4) j / 128) in fp32, rounded to bf16; [1, 256]."""
    freqs = torch.exp(-math.log(10000) * torch.arange(start=0, end=128, dtype=torch.float32) / 128)
    args = t.reshape(1, 1).float().cpu() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1).to(_act()).to(device)


def _mlp_silu(wd, a, b, x):
    return _lin(wd, b, F.silu(_lin(wd, a, x)))


def token_refiner_block(wd, j, x, c, mask, heads):
    """pre_infer.py:105-125 / 127-145: adaLN gates, LN(affine), masked self-attention over the text tokens, MLP(SiLU)."""
    p = f"txt_in.individual_token_refiner.blocks.{j}."
    g_msa, g_mlp = _lin(wd, p + "adaLN_modulation.1", F.silu(c)).chunk(2, dim=1)
    n = layer_norm(x, wd[p + "norm1.weight"], wd[p + "norm1.bias"])
    q, k, v = _split_heads(_lin(wd, p + "self_attn_qkv", n), heads)  # [L, H, D] each
    qs, ks, vs = (t.unsqueeze(0).transpose(1, 2) for t in (q, k, v))
    a = F.scaled_dot_product_attention(qs, ks, vs, attn_mask=mask).transpose(1, 2).reshape(x.shape[0], -1)
    x1 = x + _lin(wd, p + "self_attn_proj", a) * g_msa
    n2 = layer_norm(x1, wd[p + "norm2.weight"], wd[p + "norm2.bias"])
    return x1 + _mlp_silu(wd, p + "mlp.fc1", p + "mlp.fc2", n2) * g_mlp


def pre_infer(wd, dims, latents, t, guidance, text_states, text_mask, text_states_2):
    """pre_infer.py:14-60 (t2v).  latents [1,16,T,H,W]; text_states [1,L,4096] bf16; text_mask [1,L] int; text_states_2
    [1,768] bf16.  Returns img [S, hidden], txt [L, hidden], vec [1, hidden], cu_seqlens_qkv, max_seqlen_qkv."""
    dev = latents.device
    time_out = _mlp_silu(wd, "time_in.mlp.0", "time_in.mlp.2", _t_embed(t, dev))
    img = F.conv3d(latents, wd["img_in.proj.weight"], wd["img_in.proj.bias"], stride=(1, 2, 2)).flatten(2).transpose(1, 2)[0]
    # text: timestep- and context-aware conditioning vector, input embedding, two refiner blocks
    t_aware = _mlp_silu(wd, "txt_in.t_embedder.mlp.0", "txt_in.t_embedder.mlp.2", _t_embed(t, dev))
    mask_float = text_mask.float().unsqueeze(-1).to(_act())
    ctx = (text_states * mask_float).sum(dim=1) / mask_float.sum(dim=1)
    c = t_aware + _mlp_silu(wd, "txt_in.c_embedder.linear_1", "txt_in.c_embedder.linear_2", ctx)
    x = _lin(wd, "txt_in.input_embedder", text_states[0])
    L = text_mask.shape[1]
    m1 = text_mask.view(1, 1, 1, L).repeat(1, 1, L, 1)
    mask = (m1 & m1.transpose(2, 3)).bool()
    mask[:, :, :, 0] = True
    for j in range(2):
        x = token_refiner_block(wd, j, x, c, mask, dims["heads"])
    vec = time_out + _mlp_silu(wd, "vector_in.in_layer", "vector_in.out_layer", text_states_2)
    g_embed = _t_embed(guidance, dev)
    vec = vec + _mlp_silu(wd, "guidance_in.mlp.0", "guidance_in.mlp.2", g_embed)
    n_img = img.shape[0]
    s1 = int(text_mask.sum()) + n_img
    cu = torch.tensor([0, s1, L + n_img], dtype=torch.int32)
    return img, x, vec, cu, n_img + L


def post_infer(wd, img, vec, latent_shape):
    """post_infer.py:11-33: adaLN (shift, scale) → LN → modulate → fp32 Linear (MM "Default-Force-FP32") → unpatchify."""
    shift, scale = _lin(wd, "final_layer.adaLN_modulation.1", F.silu(vec)).chunk(2, dim=1)
    out = layer_norm(img) * (1 + scale) + shift
    out = mm(out.to(torch.float32), wd["final_layer.linear.weight"].float(), wd["final_layer.linear.bias"].float())
    _, _, ot, oh, ow = latent_shape
    tt, th, tw = ot, oh // 2, ow // 2
    out = out.reshape(1, tt, th, tw, 16, 1, 2, 2)
    out = torch.einsum("nthwcopq->nctohpwq", out)
    return out.reshape(1, 16, tt, th * 2, tw * 2)


def forward(wd, dims, latents, t, guidance, text_states, text_mask, text_states_2, freqs):
    """HunyuanModel.infer (model.py:151-158): pre → blocks → post; returns noise_pred fp32 [1,16,T,H,W]."""
    img, txt, vec, cu, _ = pre_infer(wd, dims, latents, t, guidance, text_states, text_mask, text_states_2)
    img, vec = transformer_infer(wd, dims, img, txt, vec, cu, freqs)
    return post_infer(wd, img, vec, latents.shape)
