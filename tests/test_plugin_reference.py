"""Drop-in check against the REAL reference (authoring container only: skipped where /root/reference is absent,
e.g. on the GPU box): our keys register into LightX2V's own registries, and the reference's weight classes build and
load their trees with our operator objects selected purely by config strings."""
import pytest
import torch

from oracle import ref_import

pytestmark = pytest.mark.skipif(not ref_import.reference_available(), reason="reference checkout not present")


def test_register_and_build_reference_weight_tree():
    ref_import.patch_and_import()
    import lightx2v_amd.plugin as plugin
    from lightx2v_amd import ops, synth
    from lightx2v.utils import registry_factory as rf

    plugin.register_into_reference()
    assert plugin.register_into_reference() == []  # idempotent
    assert rf.MM_WEIGHT_REGISTER["Hip-bf16"] is ops.MMWeightHip
    assert rf.ATTN_WEIGHT_REGISTER["hip_flash"] is ops.HipFlashAttnWeight

    dims = synth.WAN_DIMS["wan-tiny"]
    cfg = ref_import.make_config(dims, mm_config={"mm_type": "Hip-bf16"}, self_attn_1_type="hip_flash", cross_attn_1_type="hip_flash")
    from lightx2v.models.networks.wan.weights.transformer_weights import WanTransformerWeights

    tw = WanTransformerWeights(cfg)  # the reference's own class, our operators inside
    wd = synth.synth_wan_weights(dims, seed=0)
    tw.load(wd)
    blk = tw.blocks[0]
    assert isinstance(blk.compute_phases[1].self_attn_q, ops.MMWeightHip)
    assert isinstance(blk.compute_phases[1].self_attn_1, ops.HipFlashAttnWeight)
    assert blk.compute_phases[3].ffn_0.weight.shape == (dims["ffn_dim"], dims["dim"])  # checkpoint [N,K] layout kept
    sd = tw.state_dict()
    assert torch.equal(sd["blocks.0.ffn.0.weight"], wd["blocks.0.ffn.0.weight"])


def test_fused_driver_hook_and_same_signatures():
    ref_import.patch_and_import()
    import inspect

    import lightx2v_amd.plugin as plugin
    from lightx2v_amd import wan
    from lightx2v.models.networks.wan.infer.transformer_infer import WanTransformerInfer as RefInfer

    for name in ("infer", "infer_block", "infer_modulation", "_infer_without_offload"):
        ref_params = list(inspect.signature(getattr(RefInfer, name)).parameters)
        our_params = list(inspect.signature(getattr(wan.WanTransformerInfer, name)).parameters)
        assert ref_params == our_params, (name, ref_params, our_params)
    plugin.use_fused_wan_block()


@pytest.mark.gpu
def test_reference_weight_tree_runs_our_operators_on_the_gpu():
    """With a reference checkout next to a GPU (`X2V_REFERENCE_ROOT=/path/to/LightX2V pytest -m gpu tests/test_plugin_reference.py`): the
    reference's own `WanTransformerWeights` tree, built from config strings only, moved to the device with its own `to_cuda`, runs the HIP
    operators — every distinct operator kind of block 0 is applied on the device and compared with the oracle's op (tolerances of
    test_gpu_ops.py).  Skipped on the GPU box of the build (no reference there) and on GPU-less hosts."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ref_import.patch_and_import()
    import lightx2v_amd.plugin as plugin
    from lightx2v_amd import ops, synth
    from oracle import wan_oracle as O
    from tests.util import assert_bf16_close

    plugin.register_into_reference()
    dims = synth.WAN_DIMS["wan-tiny"]
    cfg = ref_import.make_config(dims, mm_config={"mm_type": "Hip-bf16"}, self_attn_1_type="hip_flash", cross_attn_1_type="hip_flash")
    from lightx2v.models.networks.wan.weights.transformer_weights import WanTransformerWeights

    tw = WanTransformerWeights(cfg)
    wd = synth.synth_wan_weights(dims, seed=0)
    tw.load(wd)
    tw.to_cuda()
    ph = tw.blocks[0].compute_phases
    D, H = dims["dim"], dims["num_heads"]
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(96, D, generator=gen).to(torch.bfloat16)
    # MM (mm_weight.py:81-88)
    q_op = ph[1].self_attn_q
    assert isinstance(q_op, ops.MMWeightHip) and q_op.weight.is_cuda
    got = q_op.apply(x.cuda())
    assert_bf16_close(got, O.mm(x, wd["blocks.0.self_attn.q.weight"], wd["blocks.0.self_attn.q.bias"]), ulps=1, atol=2e-3, bad_frac=2e-3, name="reference tree: self_attn_q.apply")
    # attention through the registered weight class, the reference's call signature (attn_weight.py:229-239; transformer_infer.py:369-379)
    q, k, v = (torch.randn(96, H, D // H, generator=gen).to(torch.bfloat16) for _ in range(3))
    cu = torch.tensor([0, 96], dtype=torch.int32)
    attn = ph[1].self_attn_1.apply(q=q.cuda(), k=k.cuda(), v=v.cuda(), cu_seqlens_q=cu, cu_seqlens_kv=cu, max_seqlen_q=96, max_seqlen_kv=96, model_cls="wan2.1")
    ref = O.sdpa(q, k, v)
    err = (attn.float().cpu().reshape(96, -1) - ref.float().reshape(96, -1)).abs()
    assert float(err.max()) <= 1e-3 * float(ref.float().abs().max()) + 4e-3
