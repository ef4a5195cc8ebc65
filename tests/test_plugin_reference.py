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
