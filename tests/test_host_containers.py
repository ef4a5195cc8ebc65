"""Host-side plumbing the reference's model code is written against: the string-keyed operator registries
(lightx2v/utils/registry_factory.py) and the recursive weight containers (lightx2v/common/modules/weight_module.py).
Pure Python, no device."""
import pytest

from lightx2v_amd.registry import DuplicateKey, Registry
from lightx2v_amd.weight_module import WeightModule, WeightModuleList


def test_registry_decorator_alias_and_lookup():
    reg = Registry("demo")

    @reg("fast")
    class Fast:
        pass

    @reg
    class Plain:
        pass

    assert reg["fast"] is Fast and reg["Plain"] is Plain
    assert "fast" in reg and "slow" not in reg and len(reg) == 2
    assert sorted(reg.keys()) == ["Plain", "fast"] and dict(reg.items())["fast"] is Fast and Fast in reg.values()
    with pytest.raises(Exception):  # the reference raises a bare Exception for a taken key
        reg("fast")(Plain)
    with pytest.raises(DuplicateKey):
        reg.register(Plain)
    reg["Default"] = Fast  # item assignment is an explicit alias / override, as the reference's dict-style writes
    reg["fast"] = Plain
    assert reg["Default"] is Fast and reg["fast"] is Plain
    with pytest.raises(KeyError, match="demo registry has no 'nope'"):
        reg["nope"]
    with pytest.raises(TypeError):
        reg["bad"] = 3
    assert reg.get("nope") is None


class _Leaf:
    def __init__(self, name, log):
        self.name, self.log, self.cfg = name, log, None

    def set_config(self, cfg):
        self.cfg = cfg

    def load(self, wd):
        self.log.append(("load", self.name))
        self.value = wd[self.name]

    def state_dict(self, dest):
        dest[self.name] = self.value
        return dest

    def to_cuda(self, non_blocking=False):
        self.log.append(("cuda", self.name, non_blocking))

    def to_cpu(self, non_blocking=False):
        self.log.append(("cpu", self.name, non_blocking))

    def _calculate_size(self):
        return 10

    def clear(self):
        self.log.append(("clear", self.name))


class _Block(WeightModule):
    def __init__(self, i, log, config):
        super().__init__()
        self.config = config
        self.register_parameter("modulation", _Leaf(f"blocks.{i}.modulation", log))
        self.add_module("q", _Leaf(f"blocks.{i}.q.weight", log))
        self.add_module("absent", None)


class _Tree(WeightModule):
    def __init__(self, log):
        super().__init__()
        self.config = {"mm_config": {"mm_type": "Hip-bf16"}}
        self.register_parameter("head", _Leaf("head.weight", log))
        self.blocks = WeightModuleList([_Block(i, log, self.config) for i in range(2)])
        self.add_module("blocks", self.blocks)


def test_weight_module_tree_walks():
    log = []
    tree = _Tree(log)
    wd = {"head.weight": 1, "blocks.0.modulation": 2, "blocks.0.q.weight": 3, "blocks.1.modulation": 4, "blocks.1.q.weight": 5}
    tree.load(wd)
    # sub-modules load before parameters at every level (the reference's order); every leaf got the mm_config
    assert [n for op, n in log if op == "load"] == ["blocks.0.q.weight", "blocks.0.modulation", "blocks.1.q.weight", "blocks.1.modulation", "head.weight"]
    assert tree.head.cfg == {"mm_type": "Hip-bf16"} and tree.blocks[1].q.cfg == {"mm_type": "Hip-bf16"}
    # state_dict: parameters first, then sub-modules
    assert list(tree.state_dict()) == ["head.weight", "blocks.0.modulation", "blocks.0.q.weight", "blocks.1.modulation", "blocks.1.q.weight"]
    assert tree.state_dict() == wd
    names = [n for n, _ in tree.named_parameters()]
    assert names == ["head", "blocks.0.modulation", "blocks.0.q", "blocks.1.modulation", "blocks.1.q"]
    assert tree.calculate_size() == 10  # leaves of this level only (the reference sizes one block at a time for its offload manager)
    assert tree.blocks[0].calculate_size() == 20 and len(tree.blocks) == 2 and [b for b in tree.blocks] == [tree.blocks[0], tree.blocks[1]]
    del log[:]
    tree.blocks[0].to_cuda_async()
    tree.blocks[0].to_cpu()
    tree.blocks[0].clear()
    assert log == [("cuda", "blocks.0.q.weight", True), ("cuda", "blocks.0.modulation", True), ("cpu", "blocks.0.q.weight", False),
                   ("cpu", "blocks.0.modulation", False), ("clear", "blocks.0.q.weight"), ("clear", "blocks.0.modulation")]
    # re-attaching a name replaces the child instead of loading / exporting it twice
    replacement = _Leaf("blocks.0.q.weight", log)
    tree.blocks[0].add_module("q", replacement)
    assert tree.blocks[0].q is replacement
    tree.blocks[0].load(wd)
    assert list(tree.blocks[0].state_dict()) == ["blocks.0.modulation", "blocks.0.q.weight"]


def test_load_from_disk_reaches_every_child_that_can():
    """reference weight_module.py:37-45: the lazy-load path forwards to sub-modules and parameters that implement it."""
    log = []

    class _Lazy(_Leaf):
        def load_from_disk(self):
            self.log.append(("disk", self.name))

    root, inner = WeightModule(), WeightModule()
    root.config = inner.config = {"mm_config": {}}
    inner.add_module("a", _Lazy("a", log))
    inner.register_parameter("p", _Lazy("p", log))
    inner.add_module("plain", _Leaf("plain", log))  # no load_from_disk: skipped
    root.add_module("inner", inner)
    root.add_module("blocks", WeightModuleList([inner]))
    root.load_from_disk()
    assert log == [("disk", "a"), ("disk", "p"), ("disk", "a"), ("disk", "p")]


def test_vae_chunk_bounds_cover_every_latent_frame_once():
    """WanVAE_.decode's passes: frame 0 alone, then chunk_frames at a time, every frame exactly once and in order (the bit-identity of the
    chunkings themselves is a GPU test: tests/test_gpu_vae.py::test_vae_decode_frame_batched_is_bit_identical)."""
    from lightx2v_amd.vae import chunk_bounds

    for t in (1, 2, 3, 5, 6, 21, 33):
        for g in (1, 2, 3, 4, 8, 64):
            ch = chunk_bounds(t, g)
            assert ch[0] == (0, 1) and ch[-1][1] == t
            assert all(a < b for a, b in ch) and all(ch[i][1] == ch[i + 1][0] for i in range(len(ch) - 1))
            assert all(b - a <= g for a, b in ch[1:])
    assert chunk_bounds(21, 4) == [(0, 1), (1, 5), (5, 9), (9, 13), (13, 17), (17, 21)]


def test_cfg_form_by_size_rule():
    """The by-size choice of how one GPU runs the two CFG forwards (wan.cfg_form_by_size), at the measured shapes and the thresholds."""
    from lightx2v_amd import synth
    from lightx2v_amd.wan import cfg_form_by_size

    def form(workload):
        wl = synth.WORKLOADS[workload]
        return cfg_form_by_size(synth.seq_len_of(wl["target_shape"]), synth.WAN_DIMS[wl["model"]]["num_heads"])

    assert form("wan1.3b_480px49f") == "streams"      # 80 query blocks x 12 heads = 960 workgroups
    assert form("wan1.3b_480px81f") == "streams"      # 128 x 12 = 1536
    assert form("wan1.3b_720px81f") == "pair"         # 296 x 12 = 3552
    assert form("wan14b_480px81f") == "pair"          # 128 x 40 = 5120
    assert form("wan14b_720px81f") == "pair"          # 296 x 40 = 11 840
    assert form("wan1.3b_256x256x17f") == "streams"
    assert cfg_form_by_size(256 * 170, 12) == "streams" and cfg_form_by_size(256 * 171, 12) == "pair"  # 2040 / 2052 workgroups: the one threshold
    assert cfg_form_by_size(256 * 170 + 1, 12) == "pair"  # a ragged last block counts


def test_patch_embedding_state_dict_returns_the_checkpoint_tensor():
    """PatchEmbedConv3dHip zero-pads its [D, C*4] GEMM operand to a multiple of 64 columns at load (i2v: 36 channels x 4 = 144 -> 192);
    state_dict() must still export the checkpoint's [D, C, 1, 2, 2] tensor (conv3d.py:29-75 / weight_module.py:47-56 round trip)."""
    import torch

    from lightx2v_amd import ops

    for c in (16, 36):
        w = torch.randn(32, c, 1, 2, 2).to(torch.bfloat16)
        b = torch.randn(32).to(torch.bfloat16)
        op = ops.PatchEmbedConv3dHip("patch_embedding.weight", "patch_embedding.bias", stride=(1, 2, 2))
        op.load({"patch_embedding.weight": w, "patch_embedding.bias": b})
        assert op.weight.shape[1] % 64 == 0
        sd = op.state_dict()
        assert sd["patch_embedding.weight"].shape == w.shape and torch.equal(sd["patch_embedding.weight"], w) and torch.equal(sd["patch_embedding.bias"], b)


def test_i2v_benchmark_workloads_are_the_reference_bench_config():
    """synth.WORKLOADS' i2v entries restate configs/bench/lightx2v_2.json (I2V-14B, 81 frames, 40 steps, CFG scale 5, shift 5) at the two resolutions the
    reference publishes numbers for (docs/EN/source/getting_started/benchmark_source.md:29-57); `workload_setup` hands bench.py / tools/e2e.py the i2v checkpoint's
    tensor set (36-channel patch embedding, img_emb MLP, k_img / v_img / norm_k_img per block), the config overrides and the seeded CLIP / VAE-encode stand-ins."""
    import torch

    from lightx2v_amd import synth

    for name, hw in (("wan14b_i2v_720px81f", (90, 160)), ("wan14b_i2v_480px81f", (60, 104))):
        wl = synth.WORKLOADS[name]
        assert wl["frames"] == 81 and wl["infer_steps"] == 40 and wl["sample_guide_scale"] == 5.0 and wl["sample_shift"] == 5.0
        assert wl["target_shape"] == (16, 21, *hw) and synth.WAN_DIMS[wl["model"]]["task"] == "i2v"
    dims, overrides, wd, lat, inputs = synth.workload_setup("wan-tiny-i2v", device="cpu")
    assert overrides["task"] == "i2v" and overrides["in_dim"] == 36 and overrides["cross_attn_2_type"] == "hip_flash"
    assert wd["patch_embedding.weight"].shape[1] == 36 and "img_emb.proj.1.weight" in wd and "blocks.0.cross_attn.k_img.weight" in wd
    img = inputs["image_encoder_output"]
    assert img["clip_encoder_out"].shape == (synth.I2V_CLIP_TOKENS, dims["clip_dim"]) and img["vae_encode_out"].shape == (20, 3, 8, 8)
    assert lat.shape == (16, 3, 8, 8) and lat.dtype == torch.float32
    d2, o2, _, _, in2 = synth.workload_setup("wan-tiny", device="cpu")
    assert "task" not in o2 and "image_encoder_output" not in in2
