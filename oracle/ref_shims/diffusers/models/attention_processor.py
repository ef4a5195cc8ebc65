"""`Attention` as the reference's VAE mid block constructs it (unet_causal_3d_blocks.py:578-590): heads = 1 of dim C, GroupNorm over
channels, to_q / to_k / to_v / to_out[0] Linear with bias, residual connection, rescale_output_factor, `_from_deprecated_attn_block`.
RESTATED from the published diffusers 0.29.2 sources (attention_processor.py: Attention.__init__, prepare_attention_mask,
AttnProcessor2_0.__call__ — the default processor when torch has scaled_dot_product_attention).  Parameter names are diffusers'
(`group_norm`, `to_q`, `to_k`, `to_v`, `to_out.0`), so the released VAE checkpoint keys load."""
import torch
import torch.nn.functional as F
from torch import nn


class SpatialNorm:
    def __init__(self, *a, **k):
        raise NotImplementedError("SpatialNorm is not used by the Hunyuan VAE decode path (norm_type='group')")


class AttnProcessor2_0:
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        residual = hidden_states
        batch, seq_len, _ = hidden_states.shape
        if attention_mask is not None:
            attention_mask = attn.prepare_attention_mask(attention_mask, seq_len, batch)
            attention_mask = attention_mask.view(batch, attn.heads, -1, attention_mask.shape[-1])
        if attn.group_norm is not None:
            hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        query = attn.to_q(hidden_states)
        key = attn.to_k(hidden_states)
        value = attn.to_v(hidden_states)
        head_dim = key.shape[-1] // attn.heads
        query = query.view(batch, -1, attn.heads, head_dim).transpose(1, 2)
        key = key.view(batch, -1, attn.heads, head_dim).transpose(1, 2)
        value = value.view(batch, -1, attn.heads, head_dim).transpose(1, 2)
        hidden_states = F.scaled_dot_product_attention(query, key, value, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        hidden_states = hidden_states.transpose(1, 2).reshape(batch, -1, attn.heads * head_dim).to(query.dtype)
        hidden_states = attn.to_out[0](hidden_states)
        hidden_states = attn.to_out[1](hidden_states)
        if attn.residual_connection:
            hidden_states = hidden_states + residual
        return hidden_states / attn.rescale_output_factor


AttnProcessor = AttnProcessor2_0
AttnAddedKVProcessor = AttnProcessor2_0
AttentionProcessor = AttnProcessor2_0
ADDED_KV_ATTENTION_PROCESSORS = ()
CROSS_ATTENTION_PROCESSORS = (AttnProcessor2_0,)


class Attention(nn.Module):
    def __init__(self, query_dim, heads=8, dim_head=64, rescale_output_factor=1.0, eps=1e-5, norm_num_groups=None, spatial_norm_dim=None,
                 residual_connection=False, bias=False, upcast_softmax=False, _from_deprecated_attn_block=False, out_bias=True, processor=None):
        super().__init__()
        if spatial_norm_dim is not None:
            raise NotImplementedError("spatial norm")
        inner = dim_head * heads
        self.heads, self.scale = heads, dim_head**-0.5
        self.rescale_output_factor, self.residual_connection, self.upcast_softmax = rescale_output_factor, residual_connection, upcast_softmax
        self.group_norm = nn.GroupNorm(num_channels=query_dim, num_groups=norm_num_groups, eps=eps, affine=True) if norm_num_groups is not None else None
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=out_bias), nn.Dropout(0.0)])
        self.processor = processor or AttnProcessor2_0()

    def set_processor(self, processor):
        self.processor = processor

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        if attention_mask.shape[-1] != target_length:
            attention_mask = F.pad(attention_mask, (0, target_length), value=0.0)
        if attention_mask.shape[0] < batch_size * self.heads:
            attention_mask = attention_mask.repeat_interleave(self.heads, dim=0)
        return attention_mask

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask, **kw)
