"""Drop-in registration into an importable LightX2V (the reference) — see INTEGRATION.md.

    import lightx2v_amd.plugin as x2v
    x2v.register_into_reference()          # adds our keys to lightx2v.utils.registry_factory registries
    x2v.use_fused_wan_block()              # optional: WanModel picks the fused HIP block driver
    x2v.use_fused_hunyuan_block()          # optional: HunyuanModel picks the fused HIP double / single block driver

After that an unchanged LightX2V config selects the HIP path by string:
    "mm_config": {"mm_type": "Hip-bf16"}   (or "W-fp8-channel-sym-A-fp8-channel-sym-dynamic-Hip")
    "self_attn_1_type": "hip_flash", "cross_attn_1_type": "hip_flash", "attention_type": "hip_flash"
"""
from . import hunyuan, ops, wan


def register_into_reference():
    """reference: lightx2v/utils/registry_factory.py:47-56 — `Register.register` raises on duplicate keys, so this
    is idempotent only through the `in` check."""
    from lightx2v.utils import registry_factory as rf

    added = []
    for reg, key, cls in (
        (rf.MM_WEIGHT_REGISTER, "Hip-bf16", ops.MMWeightHip),
        (rf.MM_WEIGHT_REGISTER, "W-fp8-channel-sym-A-fp8-channel-sym-dynamic-Hip", ops.MMWeightFp8Hip),
        (rf.MM_WEIGHT_REGISTER, "W-mxfp8-A-mxfp8-dynamic-Hip", ops.MMWeightMxfp8Hip),
        (rf.ATTN_WEIGHT_REGISTER, "hip_flash", ops.HipFlashAttnWeight),
        (rf.RMS_WEIGHT_REGISTER, "hip", ops.RMSWeightHip),
        (rf.LN_WEIGHT_REGISTER, "hip", ops.LNWeightHip),
        (rf.CONV3D_WEIGHT_REGISTER, "hip_patch", ops.PatchEmbedConv3dHip),
    ):
        if key not in reg:
            reg.register(cls, key=key)
            added.append(key)
    # functional attention dispatcher (lightx2v/attentions/__init__.py:8-20) used by ulysses_attn
    import lightx2v.attentions as la

    if not getattr(la.attention, "_x2v_wrapped", False):
        _orig = la.attention

        def attention(attention_type="flash_attn2", *args, **kwargs):
            if attention_type == "hip_flash":
                return ops.hip_flash(*args, **kwargs)
            return _orig(attention_type, *args, **kwargs)

        attention._x2v_wrapped = True
        la.attention = attention
    return added


def use_fused_wan_block():
    """Make the reference's WanModel build our fused block driver (reference hook: wan/model.py:61-75 chooses
    `transformer_infer_class`).  The driver consumes the reference's own weight trees: MM objects must be the HIP
    classes (mm_type above); norm objects may be the reference's (only `.weight/.bias/.eps` are read)."""
    from lightx2v.models.networks.wan import model as ref_model

    orig = ref_model.WanModel._init_infer_class

    def _init_infer_class(self):
        orig(self)
        if self.config["feature_caching"] == "NoCaching" and not self.config.get("cpu_offload", False):
            self.transformer_infer_class = wan.WanTransformerInfer

    ref_model.WanModel._init_infer_class = _init_infer_class


def use_fused_hunyuan_block():
    """Make the reference's HunyuanModel build our fused double / single block driver (reference hooks: hunyuan/model.py:164-176
    `_init_infer_class` chooses `transformer_infer_class`; :93-104 `_init_weights` builds the trees from `transformer_weight_class`).  The driver
    has the reference's `infer` signature (hunyuan/infer/transformer_infer.py:31) and consumes the reference's own weight trees: the MM objects
    must be the HIP classes (`mm_config.mm_type: "Hip-bf16"` + `attention_type: "hip_flash"`), the RMS-norm objects may be the reference's
    (only `.weight` is read).  `feature_caching`: "NoCaching" and "Tea" are built; the other caching modes and `cpu_offload` keep the reference's
    classes.  The reference's weight tree hard-codes 20 + 40 blocks (weights/transformer_weights.py:9-10); with `double_blocks_num` /
    `single_blocks_num` in the config the tree class of this package (same module names, same tensor names) is used instead so that reduced
    models load."""
    from lightx2v.models.networks.hunyuan import model as ref_model

    if getattr(ref_model.HunyuanModel, "_x2v_fused", False):
        return
    orig_infer_class, orig_init_weights = ref_model.HunyuanModel._init_infer_class, ref_model.HunyuanModel._init_weights

    def _init_infer_class(self):
        orig_infer_class(self)
        if self.config.get("cpu_offload", False):
            return
        fc = self.config["feature_caching"]
        if fc == "NoCaching":
            self.transformer_infer_class = hunyuan.HunyuanTransformerInfer
        elif fc == "Tea":
            self.transformer_infer_class = hunyuan.HunyuanTransformerInferTeaCaching

    def _init_weights(self):
        if "double_blocks_num" in self.config or "single_blocks_num" in self.config:
            self.transformer_weight_class = hunyuan.HunyuanTransformerWeights
        orig_init_weights(self)

    ref_model.HunyuanModel._init_infer_class = _init_infer_class
    ref_model.HunyuanModel._init_weights = _init_weights
    ref_model.HunyuanModel._x2v_fused = True
