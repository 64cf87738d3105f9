"""The estimator's building blocks as stand-alone HIP op sequences, behind the reference's block classes
(speech/matcha/models/components/{decoder,transformer}.py, speech/cosyvoice/flow/decoder.py:36-85).

The hot path does not run through these: FlowEngine executes the same arithmetic in the row-tile fused kernels
(csrc/fused.hip).  They exist so that the reference's block-level API (Block1D / ResnetBlock1D / TimestepEmbedding /
FeedForward / BasicTransformerBlock and the Causal* subclasses) is a working drop-in, one launch per Linear / Conv1d /
norm, and so that every sub-block is parity-checked on its own against the reference's outputs
(tests/golden/blocks.npz, tests/test_gpu_blocks.py).  Reference layouts: conv blocks take [B, C, T], transformer blocks
[B, T, C]; inside everything is time-major [B, T, C] with fp32 residual streams.
"""
import torch

from . import ops
from ._lib import BF16, F32, TORCH_DT


def _dev(sd):
    t = next(iter(sd.values()))
    if not t.is_cuda:
        raise RuntimeError("the MI355X hot path has no CPU fallback; move the module to a ROCm device (.to('cuda'))")
    return t.device


class BlockOps:
    """Packed weights of one block + the op sequence.  `sd` uses the reference's state-dict keys."""

    def __init__(self, sd, dtype):
        self.dt, self.tdt, self.dev = dtype, TORCH_DT[dtype], _dev(sd)
        self.f = lambda k: sd[k].detach().to(self.dev, torch.float32).contiguous()
        self.sd = sd

    def new(self, *shape, f32=False):
        return torch.empty(*shape, dtype=torch.float32 if f32 else self.tdt, device=self.dev)

    # layout helpers --------------------------------------------------------------------------------------------
    def to_tm(self, x_bct, mask=None):
        """[B, C, T] fp32 -> time-major [B, T, C] compute-dtype copy of x * mask (mask [B, 1, T] or None)."""
        B, C_, T = x_bct.shape
        xf = torch.empty(B, T, C_, dtype=torch.float32, device=self.dev)
        ops.copy2d(x_bct.to(self.dev, torch.float32).contiguous(), F32, C_ * T, 1, T, xf, F32, T * C_, C_, 1, rows=T, cols=C_, batch=B)
        a = self.new(B, T, C_)
        m = None if mask is None else mask.to(self.dev, torch.float32).reshape(B, T).contiguous()
        ops.act_rows(xf, rows=B * T, C_=C_, rowmask=m, out_act=a, dtype=self.dt)
        return a, m

    def to_bct(self, x_tm, B, T, C_):
        out = torch.empty(B, C_, T, dtype=torch.float32, device=self.dev)
        ops.copy2d(x_tm, F32, T * C_, C_, 1, out, F32, C_ * T, 1, T, rows=T, cols=C_, batch=B)
        return out

    # blocks ----------------------------------------------------------------------------------------------------
    def conv_norm_mish(self, p, a, m, B, T, cin, causal, groups=None, addvec=None, out_f32=None, out_act=None):
        """Block1D / CausalBlock1D body on a = x * mask: conv k3 -> GroupNorm(groups) | LayerNorm -> Mish -> * mask."""
        w = ops.pack_conv1d(self.f(p + ".block.0.weight"), self.dt)
        cout = w.shape[0]
        c = self.new(B, T, cout, f32=True)
        ops.conv1d(a, w, T=T, Cin=cin, k=3, pad_left=(2 if causal else 1), dtype=self.dt, batch=B,
                   bias=self.f(p + ".block.0.bias"), out_f32=c)
        if causal:                                         # Sequential(CausalConv1d, Transpose, LayerNorm, Transpose, Mish)
            ops.rownorm(c, self.f(p + ".block.2.weight"), self.f(p + ".block.2.bias"), 1e-5, rows=T, C_=cout, batch=B, act="mish",
                        rowmask=m, addvec=addvec, av_bstride=(addvec.shape[-1] if addvec is not None else None),
                        out_f32=out_f32, out_act=out_act, dtype=self.dt)
        else:                                              # Sequential(Conv1d, GroupNorm, Mish)
            assert addvec is None
            y = self.new(B, T, cout, f32=True)
            ops.groupnorm(c, self.f(p + ".block.1.weight"), self.f(p + ".block.1.bias"), y, B=B, T=T, C_=cout, groups=groups,
                          dtype=F32, act="mish", rowmask=m)
            if out_f32 is not None:
                out_f32.copy_(y)
            if out_act is not None:
                ops.act_rows(y, rows=B * T, C_=cout, out_act=out_act, dtype=self.dt)
        return cout

    def block1d(self, x, mask, causal, groups=8):
        """Block1D.forward (matcha decoder.py:41-43) / CausalBlock1D.forward: [B, cin, T] -> [B, cout, T]."""
        B, cin, T = x.shape
        a, m = self.to_tm(x, mask)
        sub = BlockOps({"b." + k: v for k, v in self.sd.items()}, self.dt)
        cout = self.sd["block.0.weight"].shape[0]
        y = self.new(B, T, cout, f32=True)
        sub.conv_norm_mish("b", a, m, B, T, cin, causal, groups, out_f32=y)
        return self.to_bct(y, B, T, cout)

    def resnet(self, x, mask, time_emb, causal, groups=8):
        """ResnetBlock1D.forward (matcha decoder.py:56-61) / CausalResnetBlock1D: [B, cin, T] -> [B, cout, T]."""
        B, cin, T = x.shape
        a, m = self.to_tm(x, mask)
        cout = self.sd["res_conv.weight"].shape[0]
        te = time_emb.to(self.dev, torch.float32).contiguous()
        tm_ = self.new(B, te.shape[1])
        ops.act_rows(te, rows=B, C_=te.shape[1], act="mish", out_act=tm_, dtype=self.dt)          # mlp = Sequential(Mish, Linear)
        tv = self.new(B, cout, f32=True)
        ops.linear(tm_, ops.pack_linear(self.f("mlp.1.weight"), self.dt), te.shape[1], dtype=self.dt, bias=self.f("mlp.1.bias"), out_f32=tv)
        h1 = self.new(B, T, cout)
        if causal:
            self.conv_norm_mish("block1", a, m, B, T, cin, True, addvec=tv, out_act=h1)             # ((y*m) + t) * m
        else:
            y1 = self.new(B, T, cout, f32=True)
            self.conv_norm_mish("block1", a, m, B, T, cin, False, groups, out_f32=y1)
            y1 += tv[:, None, :]                                                                   # h += mlp(t)[..., None]
            ops.act_rows(y1, rows=B * T, C_=cout, rowmask=m, out_act=h1, dtype=self.dt)             # block2 sees h * mask
        h2 = self.new(B, T, cout, f32=True)
        self.conv_norm_mish("block2", h1, m, B, T, cout, causal, groups, out_f32=h2)
        out = self.new(B, T, cout, f32=True)
        ops.conv1d(a, ops.pack_conv1d(self.f("res_conv.weight"), self.dt), T=T, Cin=cin, k=1, dtype=self.dt, batch=B,
                   bias=self.f("res_conv.bias"), residual=h2, out_f32=out)
        return self.to_bct(out, B, T, cout)

    def timestep_embedding(self, sample):
        """TimestepEmbedding.forward (matcha decoder.py:100-117, act 'silu', no cond / post act)."""
        x = sample.to(self.dev, torch.float32).contiguous()
        B, cin = x.shape
        xa = self.new(B, cin)
        ops.act_rows(x, rows=B, C_=cin, out_act=xa, dtype=self.dt)
        w1 = ops.pack_linear(self.f("linear_1.weight"), self.dt)
        h = self.new(B, w1.shape[0])
        ops.linear(xa, w1, cin, dtype=self.dt, bias=self.f("linear_1.bias"), act="silu", out_act=h)
        w2 = ops.pack_linear(self.f("linear_2.weight"), self.dt)
        out = self.new(B, w2.shape[0], f32=True)
        ops.linear(h, w2, w1.shape[0], dtype=self.dt, bias=self.f("linear_2.bias"), out_f32=out)
        return out

    def feed_forward(self, hs, prefix=""):
        """FeedForward.forward with activation_fn='gelu' (transformer.py:83-134: GELU proj -> Dropout -> Linear)."""
        x = hs.to(self.dev, torch.float32).contiguous()
        shp = x.shape
        C_ = shp[-1]
        rows = x.numel() // C_
        xa = self.new(rows, C_)
        ops.act_rows(x, rows=rows, C_=C_, out_act=xa, dtype=self.dt)
        w1 = ops.pack_linear(self.f(prefix + "net.0.proj.weight"), self.dt)
        h = self.new(rows, w1.shape[0])
        ops.linear(xa, w1, C_, dtype=self.dt, bias=self.f(prefix + "net.0.proj.bias"), act="gelu", out_act=h)
        w2 = ops.pack_linear(self.f(prefix + "net.2.weight"), self.dt)
        out = self.new(rows, w2.shape[0], f32=True)
        ops.linear(h, w2, w1.shape[0], dtype=self.dt, bias=self.f(prefix + "net.2.bias"), out_f32=out)
        return out.reshape(*shp[:-1], w2.shape[0])

    def transformer_block(self, hs, attention_mask, heads):
        """BasicTransformerBlock.forward (transformer.py:243-316), self-attention only, additive mask [B, T, T] (the
        bias of mask_to_bias: 0 keeps, <= -1e9 drops) or None.  Each row's visible keys must form a per-row prefix
        or a per-batch key set, which is all the estimator builds (pad and chunk masks); the bias is turned back into
        a key mask + chunk size for the attention kernels."""
        x = hs.to(self.dev, torch.float32).contiguous().clone()
        B, T, C_ = x.shape
        keymask, chunk = bias_to_keymask(attention_mask, B, T, self.dev)
        dt, f = self.dt, self.f
        inner = self.sd["attn1.to_q.weight"].shape[0]
        hn = self.new(B, T, C_)
        ops.rownorm(x, f("norm1.weight"), f("norm1.bias"), 1e-5, rows=T, C_=C_, batch=B, out_act=hn, dtype=dt)
        wqkv = ops.pack_linear(torch.cat([f("attn1.to_q.weight"), f("attn1.to_k.weight"), f("attn1.to_v.weight")], 0), dt)
        qkv = self.new(B, T, 3 * inner)
        ops.linear(hn, wqkv, C_, dtype=dt, out_act=qkv)
        ao = self.new(B, T, inner)
        ops.attn_dense(qkv, qkv[:, :, inner:], qkv[:, :, 2 * inner:], ao, B=B, H=heads, Tq=T, Tk=T, ldq=3 * inner, ldk=3 * inner,
                       ldv=3 * inner, ldo=inner, q_bs=T * 3 * inner, k_bs=T * 3 * inner, v_bs=T * 3 * inner, o_bs=T * inner,
                       scale=0.125, dtype=dt, keymask=keymask, chunk=chunk)
        x2 = self.new(B, T, C_, f32=True)
        ops.linear(ao, ops.pack_linear(f("attn1.to_out.0.weight"), dt), inner, dtype=dt, bias=f("attn1.to_out.0.bias"),
                   residual=x, out_f32=x2)
        ops.rownorm(x2, f("norm3.weight"), f("norm3.bias"), 1e-5, rows=T, C_=C_, batch=B, out_act=hn, dtype=dt)
        w1 = ops.pack_linear(f("ff.net.0.proj.weight"), dt)
        ff = self.new(B, T, w1.shape[0])
        ops.linear(hn, w1, C_, dtype=dt, bias=f("ff.net.0.proj.bias"), act="gelu", out_act=ff)
        out = self.new(B, T, C_, f32=True)
        ops.linear(ff, ops.pack_linear(f("ff.net.2.weight"), dt), w1.shape[0], dtype=dt, bias=f("ff.net.2.bias"), residual=x2,
                   out_f32=out)
        return out


def bias_to_keymask(bias, B, T, dev):
    """Additive attention bias [B, T, T] (or [B, 1, T]) -> (key mask fp32 [B, T] or None, chunk size).  Recognises what
    add_optional_chunk_mask + mask_to_bias produce (utils/mask.py:161-236, utils/common.py:160-168): a pad mask over keys,
    optionally and-ed with a static chunk mask."""
    if bias is None:
        return None, 0
    keep = (bias.to(dev) > -1e8)
    if keep.dim() == 2:
        keep = keep[:, None, :]
    keep = keep.expand(B, T, T) if keep.shape[1] == T else keep.expand(B, 1, T)
    last = keep[:, -1, :]                                   # the last row sees every non-padded key
    if keep.shape[1] == 1 or bool((keep == last[:, None, :]).all()):
        return (None if bool(last.all()) else last.float().contiguous()), 0
    # chunk mask: row i sees keys < (i // c + 1) * c: c = keys visible to row 0
    c = int(keep[0, 0].sum())
    pos = torch.arange(T, device=dev)
    want = (pos[None, :] < ((pos // c + 1) * c)[:, None])[None] & last[:, None, :]
    if not bool((keep == want).all()):
        raise NotImplementedError("attention_mask is neither a key-padding mask nor a static chunk mask")
    return (None if bool(last.all()) else last.float().contiguous()), c
