"""CPU checks of the MXFP8 oracle (oracle/mx_oracle.py) against the OCP-MX / reference-kernel rules it restates: known answers for the
round-up e8m0 scale, representability of the scaled elements, and the round-trip error bound that rule implies."""
import numpy as np
import torch

from oracle import mx_oracle as MX


def test_e8m0_round_up_known_answers():
    sf = np.array([0.0, 1.0, 1.0000001, 0.75, 0.5, 2.0, 3.0, 2.0**-126, 2.0**-127, 1.5 * 2.0**-127, 2.0**100, 3.0e38], dtype=np.float32)
    want = np.array([0, 127, 128, 127, 126, 128, 129, 1, 0, 1, 227, 254], dtype=np.uint8)
    assert (MX.e8m0_ceil(sf) == want).all()


def test_quant_properties():
    gen = torch.Generator().manual_seed(3)
    x = (torch.randn(37, 256, generator=gen) * torch.logspace(-6, 6, 37).unsqueeze(1)).to(torch.bfloat16)
    x[5, 32:64] = 0  # an all-zero block
    x[6, 0] = 448.0  # block max exactly representable: scale 2^0, element 448
    x[6, 1:32] = 1.0
    q, sc = MX.quant_mxfp8(x)
    assert q.dtype == torch.float8_e4m3fn and sc.shape == (37, 8) and sc.dtype == torch.uint8
    assert sc[5, 1].item() == 0 and (q[5, 32:64].float() == 0).all()
    assert sc[6, 0].item() == 127 and q[6, 0].float().item() == 448.0
    qf = q.float()
    assert torch.isfinite(qf).all() and qf.abs().max() <= 448
    # the block maximum lands in (224, 448]: the scale is the smallest power of two that keeps it representable
    blk = qf.reshape(37, 8, 32).abs().amax(-1)
    nz = x.float().reshape(37, 8, 32).abs().amax(-1) > 0
    assert (blk[nz] > 224 - 1e-3).all()
    # round trip: elementwise error <= half an e4m3 step at the element's magnitude (relative 2^-4) or half the smallest step of the block
    deq = MX.dequant(q, sc).float()
    xf = x.float()
    step = torch.from_numpy(np.ldexp(1.0, sc.numpy().astype(np.int64) - 127)).float().repeat_interleave(32, dim=1)  # 2^e per element
    assert ((deq - xf).abs() <= torch.maximum(xf.abs() * 2.0**-4, step * 2.0**-10) * (1 + 1e-6)).all()


def test_gemm_oracle_linear_in_alpha_and_matches_plain_matmul_on_exact_inputs():
    gen = torch.Generator().manual_seed(4)
    # integers up to 8 are exact in e4m3 and a power-of-two block max makes the quantisation lossless
    a = torch.randint(-8, 9, (16, 128), generator=gen).to(torch.bfloat16)
    b = torch.randint(-8, 9, (24, 128), generator=gen).to(torch.bfloat16)
    qa, sa = MX.quant_mxfp8(a)
    qb, sb = MX.quant_mxfp8(b)
    assert torch.equal(MX.dequant(qa, sa).float(), a.float()) and torch.equal(MX.dequant(qb, sb).float(), b.float())
    y = MX.gemm_mxfp8(qa, sa, qb, sb)
    assert torch.equal(y.float(), (a.float() @ b.float().T).to(torch.bfloat16).float())
    bias = torch.randn(24, generator=gen).to(torch.bfloat16)
    y2 = MX.gemm_mxfp8(qa, sa, qb, sb, alpha=0.5, bias=bias)
    assert torch.equal(y2.float(), (0.5 * (a.double() @ b.double().T) + bias.double()).to(torch.bfloat16).float())
