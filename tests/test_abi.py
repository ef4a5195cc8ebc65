"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol that
include/x2v.h declares (and the ctypes table mirrors the header), and fails loudly — never silently falls back —
when no gfx950 device is present."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "x2v.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(x2v_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from lightx2v_amd import lib

    names = _declared()
    assert len(names) >= 15
    so = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(so, n), f"{n} declared in include/x2v.h but not exported by {lib.LIB_PATH}"
    assert sorted(lib.PROTOTYPES) == names, "lightx2v_amd/lib.py PROTOTYPES and include/x2v.h disagree"
    assert "gfx950" in lib.version()


def test_ctypes_table_matches_header_signatures():
    """Every PROTOTYPES entry has the header's parameter count and kinds (pointer / 64-bit int / int / float): a drift between
    include/x2v.h and the ctypes binding would pass garbage to a kernel instead of raising."""
    from lightx2v_amd import lib

    src = open(os.path.join(ROOT, "include", "x2v.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = dict(re.findall(r"\b(x2v_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src))
    assert sorted(decls) == sorted(lib.PROTOTYPES)

    def kind(param):
        param = param.strip()
        if "*" in param:
            return "ptr"
        base = param.rsplit(" ", 1)[0].replace("const", "").strip()
        return {"int64_t": "i64", "int": "i32", "float": "f32"}[base]

    ctypes_kind = {ctypes.c_void_p: "ptr", ctypes.c_char_p: "ptr", ctypes.c_int64: "i64", ctypes.c_int: "i32", ctypes.c_float: "f32"}
    for name, params in decls.items():
        want = [] if params.strip() in ("", "void") else [kind(p) for p in params.split(",")]
        got = [ctypes_kind.get(t, "ptr") for t in lib.PROTOTYPES[name]]  # POINTER(c_int) etc. are pointers
        assert got == want, f"{name}: header {want} vs ctypes {got}"


def test_argument_validation_needs_no_gpu():
    """Shape/alignment/null checks run before any HIP call, so the error convention is testable on CPU."""
    from lightx2v_amd import lib

    L = lib._lib
    a = ctypes.c_void_p(4096)
    assert L.x2v_gemm_bf16(a, 64, a, 64, None, a, 64, 4, 64, 63, 0, None, 0, None, None) == -1  # K % 64
    assert b"K=63" in L.x2v_last_error()
    assert L.x2v_gemm_bf16(None, 64, a, 64, None, a, 64, 4, 64, 64, 0, None, 0, None, None) == -5  # null
    assert L.x2v_gemm_bf16(ctypes.c_void_p(4104), 64, a, 64, None, a, 64, 4, 64, 64, 0, None, 0, None, None) == -2  # alignment
    assert L.x2v_gemm_bf16(a, 64, a, 64, None, a, 64, 4, 64, 64, 2, None, 0, None, None) == -5  # residual epilogue without resid
    assert L.x2v_attn_fwd_bf16(a, 128, a, 128, a, 128, a, 128, 4, 4, 1, 64, 0.0, None) == -1  # head_dim != 128
    assert L.x2v_rmsnorm_bf16(a, 16, a, a, 16, 1, 12, 1e-6, 0, None) == -1
    assert L.x2v_layernorm_bf16(a, 16, None, None, a, None, a, 16, 1, 16, 1e-6, None) == -5  # scale without shift
    assert L.x2v_causal_conv3d_f32(a, None, 1, a, None, a, 1, 4, 4, 16, 16, 3, 3, 3, None) == -5  # cache frames without cache
    assert L.x2v_gemm_fp8(a, 128, a, a, 128, a, None, a, 64, 4, 64, 64, 0, None, 0, None, None) == -1  # K % 128
    # x2v_gemm_fp8_blocked(xq, ldx, x_kblock, x_kblock_stride, sx, wq, ldw, sw, bias, y, ldy, y_nblock, y_nblock_stride, M, N, K, epilogue, resid, ldr, gate, stream)
    assert L.x2v_gemm_fp8_blocked(a, 256, 64, 4096, a, a, 256, a, None, a, 64, 0, 0, 4, 64, 256, 0, None, 0, None, None) == -1  # K block of 64 e4m3 (not k * 128)
    assert b"K-block" in L.x2v_last_error()
    assert L.x2v_gemm_fp8_blocked(a, 256, 0, 0, a, a, 256, a, None, a, 32, 32, 4096, 4, 64, 256, 2, a, 64, None, None) == -5  # residual epilogue with an N-blocked y
    assert L.x2v_gemm_fp8_blocked(a, 256, 0, 0, None, a, 256, a, None, a, 64, 0, 0, 4, 64, 256, 0, None, 0, None, None) == -5  # null sx
    # the forced continuous form of the w8a8 GEMM refuses shapes it does not take (odd number of K tiles) before any launch
    assert L.x2v_gemm_fp8_variant(a, 384, a, a, 384, a, None, a, 256, 300, 256, 384, 0, None, 0, None, 5, None) == -1
    assert b"continuous" in L.x2v_last_error()
    assert L.x2v_gemm_fp8_variant(a, 512, a, a, 512, a, None, a, 256, 300, 256, 512, 0, None, 0, None, 4, None) == -5  # 3 / 4 are bf16 only


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
    from lightx2v_amd import lib

    x = torch.zeros(8, 64, dtype=torch.bfloat16)
    w = torch.zeros(64, 64, dtype=torch.bfloat16)
    with pytest.raises(lib.X2VError):
        lib.gemm(x, w)
    with pytest.raises(lib.X2VError):
        lib.rmsnorm(x, w[0])
    xq, wq = torch.zeros(8, 256, dtype=torch.float8_e4m3fn), torch.zeros(256, 256, dtype=torch.float8_e4m3fn)
    with pytest.raises(lib.X2VError):
        lib.gemm_fp8(xq, torch.ones(8, 1), wq, torch.ones(256, 1))
    with pytest.raises(lib.X2VError):
        lib.gemm_fp8_blocked(xq.view(8, 2, 128).transpose(0, 1).contiguous(), torch.ones(8, 1), wq, torch.ones(256, 1))


def test_missing_library_is_an_error(tmp_path):
    from lightx2v_amd import lib

    with pytest.raises(lib.X2VError):
        lib.load_library(str(tmp_path / "nope.so"))


def test_registry_protocol_matches_reference_keys():
    """Same registry objects/keys protocol as lightx2v/utils/registry_factory.py (duplicate key raises)."""
    from lightx2v_amd import ops, registry

    assert "Hip-bf16" in registry.MM_WEIGHT_REGISTER and "hip_flash" in registry.ATTN_WEIGHT_REGISTER
    assert "hip" in registry.RMS_WEIGHT_REGISTER and "hip" in registry.LN_WEIGHT_REGISTER
    with pytest.raises(Exception):
        registry.MM_WEIGHT_REGISTER("Hip-bf16")(ops.MMWeightHip)
    mm = registry.MM_WEIGHT_REGISTER["Hip-bf16"]("a.weight", "a.bias")
    wd = {"a.weight": torch.zeros(8, 64, dtype=torch.bfloat16), "a.bias": torch.zeros(8, dtype=torch.bfloat16)}
    mm.set_config({})
    mm.load(wd)
    sd = mm.state_dict()
    assert set(sd) == {"a.weight", "a.bias"} and sd["a.weight"].shape == (8, 64)
    assert mm._calculate_size() == 8 * 64 * 2 + 16


def test_attention_launch_plan_for_the_benchmark_shapes():
    """x2v_attn_vt_launch_plan is host arithmetic: the launch forms the benchmark's shapes take are pinned here without a GPU.  40 heads at 720p
    (single GPU, and the CFG pair launch): plain grid (+ stagger when asked); what an 8-GPU Ulysses rank launches (5 heads x 75 600 keys, either
    head->seq piece of 37 800 query rows: 148 x 5 = 740 workgroups, 740 % 8 = 4) and HunyuanVideo's rank (3 heads x 119 000 keys): XCD-aware
    mapping, no stagger; Wan-1.3B 480p: remap on; a launch below 512 workgroups: plain."""
    from lightx2v_amd import lib

    if os.environ.get("X2V_ATTN_MAP") or os.environ.get("X2V_ATTN_ROT"):
        pytest.skip("A/B override set")
    assert lib.attn_vt_launch_plan(75600, 75600, 40, stagger=True) == (False, True)
    assert lib.attn_vt_launch_plan(75648, 75600, 40, batch=2, stagger=True) == (False, True)
    assert lib.attn_vt_launch_plan(37800, 75600, 5) == (True, False)
    assert lib.attn_vt_launch_plan(18900, 75600, 10) == (False, False)  # 4-GPU rank, one of two pieces: 8 heads in flight x 38.7 MB > the cache
    assert lib.attn_vt_launch_plan(59528, 119000, 3) == (True, False)
    assert lib.attn_vt_launch_plan(20280, 20280, 12, stagger=True) == (True, True)
    assert lib.attn_vt_launch_plan(2560, 2560, 5) == (False, False)
    # cross-attention over the 512-token text context (transformer_infer.py:424-455): the persistent short-walk form (bit 9) for 4..32 whole key tiles
    # and >= 512 query blocks x heads; the 257 CLIP tokens of i2v (a partial tile), a 1182-tile self-attention walk and small launches keep one-walk workgroups
    assert lib.attn_vt_launch_plan(75600, 512, 40, with_short=True) == (False, False, True)
    assert lib.attn_vt_launch_plan(75648, 512, 40, batch=2, with_short=True) == (False, False, True)
    assert lib.attn_vt_launch_plan(20280, 512, 12, with_short=True) == (False, False, True)
    assert lib.attn_vt_launch_plan(9450, 512, 40, with_short=True) == (False, False, True)
    assert lib.attn_vt_launch_plan(75600, 512, 40, one_walk=True, with_short=True)[2] is False
    assert lib.attn_vt_launch_plan(75600, 257, 40, with_short=True)[2] is False
    assert lib.attn_vt_launch_plan(75600, 192, 40, with_short=True)[2] is False
    assert lib.attn_vt_launch_plan(75648, 75648, 40, with_short=True)[2] is False
    assert lib.attn_vt_launch_plan(2048, 512, 12, with_short=True)[2] is False
    with pytest.raises(lib.X2VError):
        lib.attn_vt_launch_plan(0, 10, 1)


def test_switches_and_gemm_kernel_form_are_reported_host_side():
    """x2v_switches / the form bit of x2v_gemm_kernel_choice are host arithmetic (no GPU): the effective process-wide A/B switches with their
    defaults, and which FORM of a tile family variant 0 runs for the benchmark's shapes — what bench.py prints into its JSON line and what the parity
    tests assert, so that a stray X2V_* variable on a rank cannot silently change kernels."""
    from lightx2v_amd import lib

    sw = lib.switches()
    assert set(sw) == {"X2V_GEMM_CONTINUOUS", "X2V_GEMM_FP8_CONTINUOUS", "X2V_ATTN_MAP", "X2V_ATTN_ROT"}
    if not any(k in os.environ for k in sw):
        assert sw == {"X2V_GEMM_CONTINUOUS": 1, "X2V_GEMM_FP8_CONTINUOUS": 2, "X2V_ATTN_MAP": -1, "X2V_ATTN_ROT": -1}
        assert lib.gemm_kernel_choice(151200, 5120, 5120, with_form=True) == (3, True)
        assert lib.gemm_kernel_choice(151200, 13824, 5120, with_form=True) == (3, True)
        assert lib.gemm_kernel_choice(75600, 5120, 5120 + 64, ldx=5184, ldw=5184, with_form=True) == (3, False)  # 81 K tiles: odd -> one tile per workgroup
        assert lib.gemm_kernel_choice(75600, 5120, 13824, fp8=True, with_form=True) == (2, True)
        assert lib.gemm_kernel_choice(75600, 5120, 5120 + 128, ldx=5248, ldw=5248, fp8=True, with_form=True) == (2, False)  # 41 K tiles -> ping-pong
        assert lib.gemm_kernel_choice(64, 5120, 5120, with_form=True) == (1, False)
    assert lib.gemm_kernel_choice(151200, 5120, 5120) == 3 and lib.gemm_kernel_choice(75600, 5120, 5120, fp8=True) == 2  # the family code alone, as before
    L = lib._lib
    assert L.x2v_switches(None, 0) == -5
    a = ctypes.c_void_p(4096)
    # x2v_quant_fp8_rowwise_blocked(x, ldx, x_kblock, x_kblock_stride, xq, ldq, scale, M, K, stream): block must divide K and be a multiple of 8
    assert L.x2v_quant_fp8_rowwise_blocked(a, 256, 100, 4096, a, 512, a, 4, 512, None) == -1
    assert L.x2v_quant_fp8_rowwise_blocked(a, 64, 128, 4096, a, 512, a, 4, 512, None) == -1  # ldx < block
    assert L.x2v_quant_fp8_rowwise_blocked(a, 256, 128, 4096, a, 512, a, 0, 512, None) == 0  # no rows: nothing launched
