"""Drop-in subset of speech/matcha/models/components/transformer.py: FeedForward (:83-134) and BasicTransformerBlock
(:138-316) as the CosyVoice2 estimator instantiates them (flow/decoder.py:349-359: self-attention only, 'gelu',
LayerNorm, no cross attention).  The reference builds them from diffusers 0.29 `Attention` / `GELU`; here torch
containers with the same parameter names hold the weights and `forward` runs on the HIP kernels (mmx/blocks.py)."""
from typing import Optional

import torch
from torch import nn

from ... import _paths  # noqa: F401
from .decoder import BlockHost


class _GELUProj(nn.Module):            # diffusers.models.attention.GELU: .proj Linear + exact gelu
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)


class FeedForward(BlockHost):
    def __init__(self, dim: int, dim_out: Optional[int] = None, mult: int = 4, dropout: float = 0.0,
                 activation_fn: str = "geglu", final_dropout: bool = False):
        super().__init__()
        if activation_fn != "gelu":
            raise NotImplementedError("activation_fn='gelu' (flow/decoder.py:356) is the instantiation on the hot path")
        inner = int(dim * mult)
        self.net = nn.ModuleList([_GELUProj(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out if dim_out is not None else dim)])

    @torch.inference_mode()
    def forward(self, hidden_states):
        return self._ops().feed_forward(hidden_states)


class _Attention(nn.Module):           # parameter names of diffusers Attention (to_q/to_k/to_v no bias, to_out.0 bias)
    def __init__(self, query_dim, heads, dim_head, dropout=0.0, bias=False):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])


class BasicTransformerBlock(BlockHost):
    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int, dropout=0.0,
                 cross_attention_dim: Optional[int] = None, activation_fn: str = "geglu", num_embeds_ada_norm: Optional[int] = None,
                 attention_bias: bool = False, only_cross_attention: bool = False, double_self_attention: bool = False,
                 upcast_attention: bool = False, norm_elementwise_affine: bool = True, norm_type: str = "layer_norm",
                 final_dropout: bool = False):
        super().__init__()
        if (cross_attention_dim is not None or only_cross_attention or double_self_attention or norm_type != "layer_norm"
                or attention_head_dim != 64 or attention_bias or activation_fn != "gelu"):
            raise NotImplementedError("self-attention, LayerNorm, 64-d heads, 'gelu' (flow/decoder.py:349-359) only")
        self.heads = num_attention_heads
        self.norm1 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine)
        self.attn1 = _Attention(dim, num_attention_heads, attention_head_dim, dropout, attention_bias)
        self.norm2, self.attn2 = None, None
        self.norm3 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine)
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn, final_dropout=final_dropout)

    @torch.inference_mode()
    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                timestep=None, cross_attention_kwargs=None, class_labels=None):
        """hidden_states [B, T, dim]; attention_mask additive [B, T, T] (mask_to_bias) or None -> [B, T, dim]
        (transformer.py:243-316: x + attn1(norm1(x)); x + ff(norm3(x)))."""
        assert encoder_hidden_states is None
        return self._ops().transformer_block(hidden_states, attention_mask, self.heads)
