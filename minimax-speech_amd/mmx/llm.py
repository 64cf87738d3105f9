"""Autoregressive speech-token LM engine (Qwen2-0.5B-shaped backbone + CosyVoice2 heads) on libmmx_hip kernels.

Reference (speech/): cosyvoice/llm/llm.py:676-711 (Qwen2LM.inference), :745-760 (inference_wrapper AR loop),
:359-371 (forward_one_step -> HF Qwen2ForCausalLM hidden_states[-1]), :259-274 (sampling_ids),
cosyvoice/utils/common.py:111-139 (ras/nucleus/random sampling).

One decode step for a batch of B sequences is 6 launches per layer (RMSNorm-folded QKV projection, RoPE + paged KV
append, paged GQA attention, o_proj + residual, RMSNorm-folded gate/up + SwiGLU, down + residual) + logits + the
device-resident sampler; it is recorded once into a hipGraph and replayed per token, with all loop state
(positions, step counters, token history, next input embedding, finished flags) on the device — the host only
polls `finished` every few steps (the reference syncs on .item() every token, llm.py:752).
Attention is full causal over the KV cache (SURVEY.md §7 "version-drift trap").
"""
import math
from typing import Dict, List, Optional

import torch

from . import ops
from ._lib import BF16, F32, H2, TORCH_DT, WEIGHT_DT, X3, X3W, is_split
from .flow import Graphed

ST_POS, ST_STEP, ST_NOUT, ST_FIN, ST_MINLEN, ST_MAXLEN, ST_SEQ, ST_ERR = range(8)


class PageAllocator:
    """Free list of physical KV pages (the vLLM seam of the reference, cli/model.py:274-283 / llm.py:715-743, keeps its KV
    cache in such pages).  A sequence holds pages for the rows it has written plus a look-ahead; pages return to the list
    when it finishes, and a queued request is admitted into the freed slot (LlmEngine.admit)."""

    def __init__(self, n_pages: int):
        self.free_pages = list(range(n_pages - 1, -1, -1))
        self.n_pages = n_pages

    @property
    def n_free(self) -> int:
        return len(self.free_pages)

    def alloc(self, n: int) -> List[int]:
        if n > len(self.free_pages):
            raise RuntimeError(f"KV cache exhausted: {n} pages wanted, {len(self.free_pages)} free of {self.n_pages}")
        out = self.free_pages[-n:][::-1]
        del self.free_pages[-n:]
        return out

    def free(self, pages: List[int]):
        self.free_pages.extend(reversed(pages))


class LlmEngine:
    # decode attention: batches of at least this many sequences use the GQA-shared kernel (one workgroup per kv head serving
    # its 7 query heads), smaller ones the per-head kernel (more workgroups for the few sequences there are)
    gqa_min_batch = 1 << 30
    use_v2 = True            # bf16 / split builds: decode step on csrc/decode.hip (False: the round-2 projection kernel)
    # split build, decode step: "f16x2" = activations as two fp16 planes (22 bits) and fp16 weights (csrc/decode.hip MMX_H2 / H2W:
    # two thirds of the activation bytes and of the MFMAs of the three-bf16-plane form, 4 instead of 6 bytes per weight on an
    # fp32 checkpoint); "bf16x3" = three bf16 planes (MMX_X3 / X3W)
    lm_planes = "f16x2"
    # weight prefetch: a side stream of the (captured) decode step touches layer l + 1's packed weights while layer l computes
    # (csrc/elementwise.hip prefetch_kernel), so that the next projections read them from the Infinity Cache; 0 = off, else the
    # number of workgroups of the prefetch launch
    prefetch = 0

    def __init__(self, sd: Dict[str, torch.Tensor], dtype=BF16, device="cuda", max_batch=1, max_ctx=2048, page=16,
                 heads=14, kv_heads=2, head_dim=64, rope_theta=1e6, eps=1e-6, speech_token_size=6561, use_graphs=True,
                 prefix="llm.model.model", share_from=None, kv_pages=None, wplanes=False, lm_planes=None):
        """wplanes (split build X3 only): every projection weight is carried as THREE bf16 planes hi + mid + lo = the checkpoint's
        fp32 value (MMX_X3W: csrc/decode.hip for the decode step, csrc/gemm.hip for the prompt pass) instead of being rounded to
        bf16 - for checkpoints whose weights are not bf16-representable (the reference loads an fp32 llm.pt, cli/model.py:67-75).
        Costs 3 x the weight bytes and 2 x the MFMAs of the plain split build."""
        self.dtype, self.tdt, self.dev = dtype, TORCH_DT[dtype], torch.device(device)
        self.wplanes = (dtype == X3 and ops.resolve_wplanes(wplanes, (v for k, v in sd.items() if v.dim() >= 2 and ("proj" in k or k == "llm_decoder.weight")))) \
            if share_from is None else share_from.wplanes
        planes = (lm_planes or self.lm_planes) if share_from is None else ("f16x2" if share_from.h2 else "bf16x3")
        assert planes in ("f16x2", "bf16x3")
        self.h2 = dtype == X3 and planes == "f16x2" and self.use_v2
        self.ddt = H2 if self.h2 else dtype               # dtype code of the decode-step kernels (csrc/decode.hip)
        if self.wplanes and not self.h2:                  # three bf16 weight planes in registers: one output tile per workgroup
            self.v2_cfg = dict(qkv=(1, 1), o=(1, 1), gu=(1, 1), down=(1, 8), head=(1, 1))
        self.split = is_split(dtype)                      # bf16 weights, fp32 activations split inside the MFMA products
        self.Hq, self.Hkv, self.D, self.eps = heads, kv_heads, head_dim, eps
        self.page, self.use_graphs = page, use_graphs
        self.eos = speech_token_size
        self.V = speech_token_size + 3
        dt = dtype
        if share_from is not None:
            # a second engine of a different batch size over the SAME packed weights and the SAME paged KV cache
            # (used to continue a partly finished batch at a smaller, cheaper batch size: see compact_from)
            o = share_from
            for k in ("n_layers", "H", "I", "layers", "wdec", "bdec", "embed_tokens", "speech_emb", "llm_emb", "inv_freq",
                      "rope_tab", "kc", "vc", "max_pages", "max_out", "trash_page", "pf_layers", "norm_w", "pages", "unfolded"):
                setattr(self, k, getattr(o, k))
            self.B = max_batch
            self.block_table = torch.full((self.B, self.max_pages), self.trash_page, dtype=torch.int32, device=self.dev)
            self.slot_pages = [[] for _ in range(self.B)]
            self._alloc_state()
            return
        f = lambda k: sd[k].detach().to(self.dev, torch.float32).contiguous()
        c = (lambda t: t.float().contiguous()) if self.wplanes else (lambda t: t.to(WEIGHT_DT[dtype]).contiguous())
        if self.wplanes:
            dt = X3W                                      # the code the weights are packed for (ops.Planed packs)
        # The RMSNorm gain is folded into the packed weights only in the fp32 build.  The bf16 and split builds keep the
        # checkpoint's bf16 weights as they are and apply the gain to the activations: in the producer's epilogue on the
        # decode step (csrc/decode.hip), in the kernel for prompt chunks (kgamma).
        self.unfolded = dtype != F32
        ks = (lambda g: None) if self.unfolded else (lambda g: g)
        self.n_layers = len({k.split(".")[4] for k in sd if k.startswith(prefix + ".layers.")})
        self.H = sd[prefix + ".norm.weight"].shape[0]
        self.I = sd[prefix + ".layers.0.mlp.gate_proj.weight"].shape[0]
        self.layers, self.pf_layers = [], []
        for l in range(self.n_layers):
            p = f"{prefix}.layers.{l}"
            a = p + ".self_attn"
            wqkv = torch.cat([f(a + ".q_proj.weight"), f(a + ".k_proj.weight"), f(a + ".v_proj.weight")], 0)
            bqkv = torch.cat([f(a + ".q_proj.bias"), f(a + ".k_proj.bias"), f(a + ".v_proj.bias")], 0).contiguous()
            wgu = torch.cat([f(p + ".mlp.gate_proj.weight"), f(p + ".mlp.up_proj.weight")], 0)
            if max_batch >= 4 or self.wplanes or self.h2:
                # row-major copies for the batched prompt pass (_prefill_batch): many prompts at once are an ordinary
                # tall GEMM over weights read once, not max_batch passes of the weight-streaming decode kernels
                self.pf_layers.append(dict(
                    wqkv=ops.pack_linear(c(wqkv), dt), wo=ops.pack_linear(c(f(a + ".o_proj.weight")), dt),
                    wgu=ops.pack_linear(c(wgu), dt), wdown=ops.pack_linear(c(f(p + ".mlp.down_proj.weight")), dt),
                    g1=f(p + ".input_layernorm.weight"), g2=f(p + ".post_attention_layernorm.weight")))
            if self.h2:                                   # fp16 packs (w * 2^8): one plane (bf16-representable checkpoint) or hi + lo
                pk = lambda w, ih=0: ops.pack_skinny_h2(w, planes=(2 if self.wplanes else 1), interleave_half=ih)
                self.layers.append(dict(wqkv=pk(wqkv), bqkv=bqkv, wo=pk(f(a + ".o_proj.weight")), wgu=pk(wgu, self.I),
                                        wdown=pk(f(p + ".mlp.down_proj.weight")),
                                        g1=f(p + ".input_layernorm.weight"), g2=f(p + ".post_attention_layernorm.weight")))
                del wqkv, wgu
                continue
            self.layers.append(dict(
                wqkv=ops.pack_skinny(c(wqkv), dtype=dt, kscale=ks(f(p + ".input_layernorm.weight"))), bqkv=bqkv,
                wo=ops.pack_skinny(c(f(a + ".o_proj.weight")), dtype=dt),
                wgu=ops.pack_skinny(c(wgu), dtype=dt, kscale=ks(f(p + ".post_attention_layernorm.weight")), interleave_half=self.I),
                wdown=ops.pack_skinny(c(f(p + ".mlp.down_proj.weight")), dtype=dt),
                g1=f(p + ".input_layernorm.weight"), g2=f(p + ".post_attention_layernorm.weight")))
            del wqkv, wgu
        self.norm_w = f(prefix + ".norm.weight")
        self.embed_tokens = f(prefix + ".embed_tokens.weight")
        if "llm_decoder.weight" in sd:
            self.wdec = ops.pack_skinny_h2(f("llm_decoder.weight"), planes=(2 if self.wplanes else 1)) if self.h2 else \
                ops.pack_skinny(c(f("llm_decoder.weight")), dtype=dt, kscale=ks(self.norm_w))
            self.bdec = f("llm_decoder.bias")
            self.speech_emb = f("speech_embedding.weight")
            self.llm_emb = f("llm_embedding.weight")
        else:                                              # backbone only (Qwen2Encoder.forward_one_step)
            self.wdec = self.bdec = self.speech_emb = self.llm_emb = None
        # HF Qwen2RotaryEmbedding inv_freq (modeling_qwen2.py: 1 / theta^(arange(0,d,2)/d)), computed like HF in fp32
        self.inv_freq = (1.0 / (rope_theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))).to(self.dev)
        # cos/sin per position exactly as HF computes them (fp32 outer product, then cos/sin): [max_ctx][cos 32 | sin 32]
        # one capacity for the KV pages, the RoPE table, the token history and every guard: whole pages
        self.max_pages = (max_ctx + page - 1) // page
        max_ctx = self.max_pages * page
        ang = torch.arange(max_ctx, dtype=torch.float32)[:, None] * self.inv_freq.cpu()[None, :]
        self.rope_tab = torch.cat([ang.cos(), ang.sin()], dim=1).contiguous().to(self.dev)
        # paged KV cache: [layers][pages][Hkv][page][D]; pages are handed out by a free-list allocator, a slot's block
        # table row lists the pages of the sequence it currently runs (idle slots point at the scratch page)
        self.B = max_batch
        npages = self.B * self.max_pages if kv_pages is None else int(kv_pages)
        self.trash_page = npages                  # scratch page: idle slots append their (ignored) KV here
        self.kc = torch.zeros(self.n_layers, npages + 1, kv_heads, page, head_dim, dtype=self.tdt, device=self.dev)
        self.vc = torch.zeros_like(self.kc)
        self.pages = PageAllocator(npages)
        self.block_table = torch.full((self.B, self.max_pages), self.trash_page, dtype=torch.int32, device=self.dev)
        self.slot_pages = [[] for _ in range(self.B)]
        self.max_out = max_ctx
        self._alloc_state()

    def _alloc_state(self):
        B = self.B
        self.state = torch.zeros(8, B, dtype=torch.int32, device=self.dev)
        self.out_tokens = torch.zeros(B, self.max_out, dtype=torch.int32, device=self.dev)
        self.sampled = torch.full((B, self.max_out), -1, dtype=torch.int32, device=self.dev)
        self.forced, self._forced_buf = None, None
        self.x_in = torch.zeros(B, self.H, device=self.dev)         # next input embedding (written by the sampler)
        self.h = torch.zeros(B, self.H, device=self.dev)            # residual stream of the step
        # its compute-dtype copy; at batch > 8 the decode step keeps it (and every other GEMM input) in the packed
        # MFMA-fragment order of include/mmx_hip.h (whole 16-row tiles)
        self.packed = B >= 4 and not self.unfolded
        self.h_act = torch.zeros(ops.packed_rows(B), self.H, dtype=self.tdt, device=self.dev)
        self.logits = torch.zeros(B, self.V, device=self.dev)
        self.logp = torch.zeros(B, self.V, device=self.dev)
        self.want_logp = False
        self.seed = 0
        self.top_p, self.top_k, self.win_size, self.tau_r = 0.8, 25, 10, 0.1     # config.yaml:46-50
        self._decode = None
        self.reserve_ahead = 1 << 30               # start(): rows reserved past the prompt (default: the whole max_len)

    # ------------------------------------------------------------------ one transformer pass over `rows` tokens/seq
    def _layers(self, h, ha, B, rows, pos, block_table, packed=False):
        """h fp32 [B*rows, H] residual stream (in place) and ha, its compute-dtype copy (kept in sync by the
        residual epilogues: it is the A operand of the next RMSNorm-folded projection).
        pos int32 [B] device, block_table [B, max_pages].
        packed (decode step, batch > 8): ha / att / act live in the packed fragment order; the first projection
        reads the fp32 residual stream itself (row-major), so no packing pass is needed for the input embedding."""
        dt, H, I = self.dtype, self.H, self.I
        n = B * rows
        assert n <= 64 and not (packed and rows != 1)
        if self.split:
            return self._layers_split(h, B, rows, pos, block_table)
        nr = ops.packed_rows(n) if packed else n
        qkv = torch.empty(n, (self.Hq + 2 * self.Hkv) * self.D, device=self.dev)
        q = torch.empty(n, self.Hq * self.D, dtype=self.tdt, device=self.dev)
        att = torch.empty(nr, self.Hq * self.D, dtype=self.tdt, device=self.dev)
        act = torch.empty(nr, I, dtype=self.tdt, device=self.dev)
        pk = packed
        kg = (lambda g: g) if self.unfolded else (lambda g: None)     # bf16 build: gain applied to the fp32 residual stream in-kernel
        for l, w in enumerate(self.layers):
            first = (packed and l == 0) or self.unfolded
            ops.skinny_gemm(h if first else ha, w["wqkv"], B=n, K=H, N=qkv.shape[1], dtype=dt, bias=w["bqkv"], rs=True,
                            eps=self.eps, epi=0, out_f32=qkv, x_packed=pk and not first, kgamma=kg(w["g1"]))
            if rows == 1:
                ops.decode_attn(qkv, self.inv_freq, pos, self.kc[l], self.vc[l], block_table, att, B=B, Hq=self.Hq,
                                Hkv=self.Hkv, page=self.page, dtype=dt, rope_tab=self.rope_tab, out_packed=pk,
                                per_head=B < self.gqa_min_batch)
            else:
                ops.rope_kv_store(qkv, self.inv_freq, pos, q, self.kc[l], self.vc[l], block_table, B=B, rows=rows,
                                  Hq=self.Hq, Hkv=self.Hkv, page=self.page, dtype=dt)
                ops.paged_attn(q, pos, self.kc[l], self.vc[l], block_table, att, B=B, rows=rows, Hq=self.Hq,
                               Hkv=self.Hkv, page=self.page, dtype=dt)
            ops.skinny_gemm(att, w["wo"], B=n, K=self.Hq * self.D, N=H, dtype=dt, epi=2, out_f32=h, out_act=ha,
                            x_packed=pk, out_packed=pk)
            ops.skinny_gemm(h if self.unfolded else ha, w["wgu"], B=n, K=H, N=I, dtype=dt, rs=True, eps=self.eps, epi=1, out_act=act,
                            x_packed=pk, out_packed=pk, kgamma=kg(w["g2"]))
            ops.skinny_gemm(act, w["wdown"], B=n, K=I, N=H, dtype=dt, epi=2, out_f32=h, out_act=ha,
                            x_packed=pk, out_packed=pk)

    # decode-step projections of the split build (csrc/decode.hip): (output tiles per workgroup, k slices across workgroups)
    v2_cfg = dict(qkv=(1, 1), o=(1, 1), gu=(2, 1), down=(2, 8), head=(2, 1))

    def _planes(self):
        """Static buffers of the split-plane decode step (csrc/decode.hip): activation planes, sum-of-squares tables (zeroed:
        unused tile slots must read 0), partial tiles and tickets of the down projection's cross-workgroup k split."""
        if not hasattr(self, "_v2"):
            H, I, R = self.H, self.I, ops.packed_rows(self.B)
            NQ = (self.Hq + 2 * self.Hkv) * self.D
            bf = lambda K: torch.zeros((2 if self.h2 else 3) if self.split else 1, R * K, dtype=torch.bfloat16, device=self.dev)
            J = self.v2_cfg["down"][1]
            self._v2 = dict(qkv=torch.empty(self.B, NQ, device=self.dev), xs_a=bf(H), xs_b=bf(H), xs_att=bf(self.Hq * self.D), xs_act=bf(I),
                            ssq_a=torch.zeros(32, ops.SSQ_SLOTS, device=self.dev), ssq_b=torch.zeros(32, ops.SSQ_SLOTS, device=self.dev),
                            part=torch.empty(J * ((H + 15) // 16) * (R // 4) * 64, device=self.dev),
                            tickets=torch.zeros((H + 15) // 16, dtype=torch.int32, device=self.dev))
        return self._v2

    def _layers_split_decode(self, x_in, h, B, pos, block_table):
        """One decode step of the split build on split-plane activations (B <= 32 sequences): one prep launch, then 5
        launches per layer.  x_in -> h (residual stream, fp32) and the planes of h * gamma; every projection's epilogue
        writes the planes (and the RMSNorm partial sums) its consumer reads.  Leaves the planes of h * norm_w and the sums
        of squares of h in xs_a / ssq_a for the head."""
        dt, H, I, c, S = self.ddt, self.H, self.I, self.v2_cfg, self._planes()
        NQ = (self.Hq + 2 * self.Hkv) * self.D
        qkv = S["qkv"][:B]
        ops.decode_prep(x_in, S["xs_a"], S["ssq_a"], B=B, K=H, gamma=self.layers[0]["g1"], h=h, dtype=dt)
        pf = int(self.prefetch)
        if pf:
            if not hasattr(self, "_pf_side"):
                self._pf_side = torch.cuda.Stream(device=self.dev)
                self._pf_sink = torch.zeros(4, dtype=torch.int32, device=self.dev)
            cur, side = torch.cuda.current_stream(), self._pf_side
        for l, w in enumerate(self.layers):
            if pf:                                        # fork: the next layer's weights (the head's after the last layer)
                nxt = self.layers[l + 1] if l + 1 < len(self.layers) else None
                ev = torch.cuda.Event()
                ev.record(cur)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    ops.prefetch4([nxt["wqkv"], nxt["wo"], nxt["wgu"], nxt["wdown"]] if nxt is not None else [self.wdec], self._pf_sink, pf)
            g_next = self.layers[l + 1]["g1"] if l + 1 < len(self.layers) else self.norm_w
            ops.skinny2(S["xs_a"], w["wqkv"], B=B, K=H, N=NQ, dtype=dt, bias=w["bqkv"], ssq_in=S["ssq_a"], eps=self.eps, epi=0, out=qkv,
                        tiles_per_wg=c["qkv"][0])
            # (bf16 build: one plane = the packed A-fragment order the attention kernels already write)
            ops.decode_attn(qkv, self.inv_freq, pos, self.kc[l], self.vc[l], block_table, S["xs_att"], B=B, Hq=self.Hq,
                            Hkv=self.Hkv, page=self.page, dtype=self.dtype, rope_tab=self.rope_tab, per_head=(self.split or B < self.gqa_min_batch),
                            out_split=("f16" if self.h2 else self.split), out_packed=not self.split)
            ops.skinny2(S["xs_att"], w["wo"], B=B, K=self.Hq * self.D, N=H, dtype=dt, epi=2, out=h, xs_out=S["xs_b"],
                        gamma_next=w["g2"], ssq_out=S["ssq_b"], tiles_per_wg=c["o"][0])
            ops.skinny2(S["xs_b"], w["wgu"], B=B, K=H, N=I, dtype=dt, ssq_in=S["ssq_b"], eps=self.eps, epi=1, xs_out=S["xs_act"],
                        tiles_per_wg=c["gu"][0])
            ops.skinny2(S["xs_act"], w["wdown"], B=B, K=I, N=H, dtype=dt, epi=2, out=h, xs_out=S["xs_a"], gamma_next=g_next,
                        ssq_out=S["ssq_a"], tiles_per_wg=c["down"][0], ksplit=c["down"][1], part=S["part"], tickets=S["tickets"])
        if pf:                                            # join: the side stream's last launch belongs to this step
            ev = torch.cuda.Event()
            ev.record(side)
            cur.wait_event(ev)

    def _layers_split(self, h, B, rows, pos, block_table):
        """The split build of _layers: every GEMM input is the fp32 tensor itself (row-major), the RMSNorm gains ride as
        kgamma, all intermediates are fp32."""
        dt, H, I = self.dtype, self.H, self.I
        n = B * rows
        qkv = torch.empty(n, (self.Hq + 2 * self.Hkv) * self.D, device=self.dev)
        q = torch.empty(n, self.Hq * self.D, device=self.dev)
        att = torch.empty(n, self.Hq * self.D, device=self.dev)
        act = torch.empty(n, I, device=self.dev)
        for l, w in enumerate(self.layers):
            ops.skinny_gemm(h, w["wqkv"], B=n, K=H, N=qkv.shape[1], dtype=dt, bias=w["bqkv"], rs=True, eps=self.eps, epi=0,
                            out_f32=qkv, kgamma=w["g1"])
            if rows == 1:
                ops.decode_attn(qkv, self.inv_freq, pos, self.kc[l], self.vc[l], block_table, att, B=B, Hq=self.Hq,
                                Hkv=self.Hkv, page=self.page, dtype=dt, rope_tab=self.rope_tab, per_head=True)
            else:
                ops.rope_kv_store(qkv, self.inv_freq, pos, q, self.kc[l], self.vc[l], block_table, B=B, rows=rows,
                                  Hq=self.Hq, Hkv=self.Hkv, page=self.page, dtype=dt)
                ops.paged_attn(q, pos, self.kc[l], self.vc[l], block_table, att, B=B, rows=rows, Hq=self.Hq,
                               Hkv=self.Hkv, page=self.page, dtype=dt)
            ops.skinny_gemm(att, w["wo"], B=n, K=self.Hq * self.D, N=H, dtype=dt, epi=2, out_f32=h)
            ops.skinny_gemm(h, w["wgu"], B=n, K=H, N=I, dtype=dt, rs=True, eps=self.eps, epi=1, out_f32=act, kgamma=w["g2"])
            ops.skinny_gemm(act, w["wdown"], B=n, K=I, N=H, dtype=dt, epi=2, out_f32=h)

    def _tail(self, B, packed=False, planes_ready=False):
        """final RMSNorm (folded) + llm_decoder + log_softmax + sampler + loop bookkeeping for all B sequences.
        planes_ready (split build): xs_a / ssq_a already hold the planes of h * norm_w (the decode step's last projection
        wrote them); otherwise they are made from self.h first."""
        if self.unfolded and B <= 32 and self.use_v2:
            S = self._planes()
            if not planes_ready:
                ops.decode_prep(self.h, S["xs_a"], S["ssq_a"], B=B, K=self.H, gamma=self.norm_w, dtype=self.ddt)
            ops.skinny2(S["xs_a"], self.wdec, B=B, K=self.H, N=self.V, dtype=self.ddt, bias=self.bdec, ssq_in=S["ssq_a"],
                        eps=self.eps, epi=0, out=self.logits, tiles_per_wg=self.v2_cfg["head"][0])
        elif self.unfolded:
            ops.skinny_gemm(self.h, self.wdec, B=B, K=self.H, N=self.V, dtype=self.dtype, bias=self.bdec, rs=True,
                            eps=self.eps, epi=0, out_f32=self.logits, kgamma=self.norm_w)
        else:
            ops.skinny_gemm(self.h_act, self.wdec, B=B, K=self.H, N=self.V, dtype=self.dtype, bias=self.bdec, rs=True,
                            eps=self.eps, epi=0, out_f32=self.logits, x_packed=packed)
        ops.sample_step(self.logits, self.state, self.out_tokens, self.speech_emb, self.x_in, V=self.V, B=B,
                        eos_id=self.eos, seed=self.seed, top_k=self.top_k, top_p=self.top_p, win_size=self.win_size,
                        tau_r=self.tau_r, sampled=self.sampled, forced=self.forced,
                        logp_out=(self.logp if self.want_logp else None))

    def _decode_step(self):
        B = self.B
        if self.unfolded and B <= 32 and self.use_v2:
            self._layers_split_decode(self.x_in, self.h, B, self.state[ST_POS], self.block_table)
            return self._tail(B, planes_ready=True)
        self.h.copy_(self.x_in)
        if not self.packed:
            self.h_act[:B].copy_(self.x_in)
        self._layers(self.h, self.h_act, B, 1, self.state[ST_POS], self.block_table, packed=self.packed)
        self._tail(B, packed=self.packed)

    # ------------------------------------------------------------------ request setup
    def build_lm_input(self, text, prompt_text, prompt_speech_token, speaker_embed=None):
        """llm.py:691-703: [sos | embed(prompt_text ++ text) | task_id | speech_emb(prompt_speech)] fp32 [L, H];
        with speaker_embed ([1, H], inference_spk, llm.py:663): [sos | speaker_embed | text | task_id | prompt speech]."""
        tok = torch.cat([prompt_text.reshape(-1), text.reshape(-1)]).to(self.dev, torch.int64)
        s = 0 if speaker_embed is None else 1
        L = 2 + s + tok.numel() + prompt_speech_token.numel()
        x = torch.empty(L, self.H, device=self.dev)
        x[0].copy_(self.llm_emb[0])
        if s:
            x[1].copy_(speaker_embed.reshape(-1))
        ops.gather_rows(tok, self.embed_tokens, out_f32=x[1 + s:1 + s + tok.numel()], dtype=F32)
        x[1 + s + tok.numel()].copy_(self.llm_emb[1])
        if prompt_speech_token.numel():
            ops.gather_rows(prompt_speech_token.reshape(-1).to(self.dev, torch.int64), self.speech_emb,
                            out_f32=x[2 + s + tok.numel():], dtype=F32)
        return x

    def speaker_conditioning(self, sd_linear_w, sd_linear_b, emb192):
        """normalize -> spk_embed_affine_layer (llm.py:184-186 / :650-653): [1,192] -> [1,H] fp32."""
        import math
        d = emb192.shape[1]
        g = torch.full((d,), 1.0 / math.sqrt(d), device=self.dev)
        en = torch.empty(1, d, dtype=self.tdt, device=self.dev)
        ops.rownorm(emb192.to(self.dev, torch.float32).contiguous(), g, None, 1e-30, rows=1, C_=d, rms=True, out_act=en, dtype=self.dtype)
        out = torch.empty(1, self.H, device=self.dev)
        ops.linear(en, sd_linear_w, d, dtype=self.dtype, bias=sd_linear_b, out_f32=out)
        return out

    # ------------------------------------------------------------------ KV pages
    def _set_pages(self, slot: int, rows: int):
        """Makes sure slot `slot` owns pages for `rows` cache rows (allocating what is missing)."""
        need = min((rows + self.page - 1) // self.page, self.max_pages)
        have = self.slot_pages[slot]
        if need > len(have):
            new = self.pages.alloc(need - len(have))
            self.block_table[slot, len(have):need] = torch.tensor(new, dtype=torch.int32)
            have.extend(new)

    def release(self, slot: int):
        """Returns the slot's pages to the free list; the slot idles on the scratch page until the next admit."""
        if self.slot_pages[slot]:
            self.pages.free(self.slot_pages[slot])
            self.slot_pages[slot] = []
            self.block_table[slot].fill_(self.trash_page)

    def ensure_capacity(self, ahead: int, pos: Optional[List[int]] = None, active: Optional[List[int]] = None):
        """Called by the host loop between decode steps: every active slot must own pages for the next `ahead` rows
        (the loop polls every few steps, so it allocates that far ahead; a full page list raises)."""
        pos = self.state[ST_POS].tolist() if pos is None else pos
        for s_ in (range(self.B) if active is None else active):
            if self.slot_pages[s_]:
                self._set_pages(s_, pos[s_] + 1 + ahead)

    @torch.no_grad()
    def admit(self, slot: int, x: torch.Tensor, min_len: int, max_len: int, seq_id: int, ahead: Optional[int] = None):
        """Continuous batching: puts a new request into an idle slot while the other slots keep decoding.  All prompt
        rows but the last are prefetched into freshly allocated pages; the last row becomes the slot's next input, so the
        next ordinary decode step of the batch computes its logits and draws its first token (same Philox key (seed,
        seq id, step 0) as a fixed-batch start) — no separate sampling pass, nothing of the other sequences is touched.
        ahead: cache rows reserved past the prompt.  None reserves the whole max_len (what start() does); a caller that
        passes less MUST call ensure_capacity() between decode steps (run_queue does) — rows past the reservation map to
        the shared scratch page."""
        L = x.shape[0]
        if L + max_len > self.max_pages * self.page:
            raise RuntimeError("sequence exceeds the KV cache")
        self.release(slot)
        self._set_pages(slot, L + (max_len if ahead is None else min(ahead, max_len)))
        x = x.to(self.dev, torch.float32).contiguous()
        for c0 in range(0, L - 1, 64):
            self._prefill_chunk(x[c0:min(L - 1, c0 + 64)], c0, slot)
        self.x_in[slot].copy_(x[L - 1])
        st = torch.tensor([L - 1, 0, 0, 0, min_len, max_len, seq_id, 0], dtype=torch.int32)
        self.state[:, slot].copy_(st)
        self.sampled[slot].fill_(-1)

    def start(self, lm_inputs: List[torch.Tensor], min_lens: List[int], max_lens: List[int], seed=0, seq_ids=None,
              forced: Optional[torch.Tensor] = None, want_logp=False):
        """Prefills every sequence (prompt rows in chunks of <= 64 through the same kernels as decode) and samples
        the first token of each.  After this, call step()/run()."""
        B = self.B
        assert len(lm_inputs) == B
        self.seed, self.want_logp = int(seed), want_logp
        if forced is not None:
            if self._forced_buf is None:
                self._forced_buf = torch.zeros(B, self.max_out, dtype=torch.int32, device=self.dev)
            self._forced_buf[:, :forced.shape[1]].copy_(forced.to(torch.int32))
            self.forced = self._forced_buf
        else:
            self.forced = None
        st = torch.zeros(8, B, dtype=torch.int32)
        for b in range(B):
            st[ST_MINLEN, b], st[ST_MAXLEN, b] = min_lens[b], max_lens[b]
            st[ST_SEQ, b] = b if seq_ids is None else seq_ids[b]
        self.state.copy_(st)
        self.sampled.fill_(-1)
        for b, x in enumerate(lm_inputs):
            assert x.shape[0] + max_lens[b] <= self.max_pages * self.page, "sequence exceeds the KV cache"
            self.release(b)
            self._set_pages(b, x.shape[0] + min(max_lens[b], self.reserve_ahead))
        if (B >= 4 or self.wplanes or self.h2) and self.pf_layers:
            self._prefill_batch(lm_inputs)
            lm_inputs = []
        for b, x in enumerate(lm_inputs):
            L = x.shape[0]
            x = x.to(self.dev, torch.float32).contiguous()
            for c0 in range(0, L, 64):
                c1 = min(L, c0 + 64)
                hc, hca = self._prefill_chunk(x[c0:c1], c0, b)
            self.h[b].copy_(hc[-1])
            self.h_act[b].copy_(hca[-1])
            self.state[ST_POS, b] = L - 1                # the sampler's +1 makes it L (= rows in the cache)
        self._tail(B)
        if self._decode is None:
            self._decode = Graphed(self._decode_step, self.use_graphs)
        elif self._graph_key != (self.forced is None, want_logp, self.seed):
            self._decode.release()
            self._decode = Graphed(self._decode_step, self.use_graphs)      # baked arguments changed: re-record
        self._graph_key = (self.forced is None, want_logp, self.seed)

    def _prefill_batch(self, lm_inputs):
        """All prompts in one pass per layer: rows = B x Lmax (shorter prompts are zero padded; a padded row only adds
        cache entries past its prompt, which the decode steps overwrite before they are read).  RMSNorm, projections
        on the windowed GEMM (weights read once for every prompt), RoPE + KV store + causal attention over the paged
        cache, SwiGLU: 9 launches per layer for the whole batch (32 prompts of 50 rows through the decode kernels cost
        ~58 ms of GPU time: the 1 GB of weights was streamed once per prompt)."""
        B, dt, H, I = self.B, self.dtype, self.H, self.I
        Ls = [int(x.shape[0]) for x in lm_inputs]
        Lm = max(Ls)
        R = B * Lm
        h = torch.zeros(B, Lm, H, device=self.dev)
        for b, x in enumerate(lm_inputs):
            h[b, :Ls[b]].copy_(x)
        h = h.reshape(R, H)
        self._gemm_layers(h, B, Lm, torch.zeros(B, dtype=torch.int32, device=self.dev), self.block_table)
        last = torch.tensor([b * Lm + Ls[b] - 1 for b in range(B)], dtype=torch.long, device=self.dev)
        hl = h.index_select(0, last)
        self.h[:B].copy_(hl)
        self.h_act[:B].copy_(hl)
        self.state[ST_POS].copy_(torch.tensor([L - 1 for L in Ls], dtype=torch.int32))

    def _gemm_layers(self, h, B, Lm, pos, block_table):
        """The layers over B x Lm prompt rows h [B * Lm, H] (fp32, in place) on the windowed GEMM: RMSNorm, projections (weights read
        once for all rows; weight planes when the packs are ops.Planed), RoPE + KV store + causal attention over the paged cache
        from position pos[b], SwiGLU - 9 launches per layer."""
        dt, H, I = self.dtype, self.H, self.I
        R = B * Lm
        a = torch.empty(R, H, dtype=self.tdt, device=self.dev)
        qkv = torch.empty(R, (self.Hq + 2 * self.Hkv) * self.D, device=self.dev)
        q = torch.empty(R, self.Hq * self.D, dtype=self.tdt, device=self.dev)
        att = torch.empty(R, self.Hq * self.D, dtype=self.tdt, device=self.dev)
        gu = torch.empty(R, 2 * I, device=self.dev)
        act = torch.empty(R, I, dtype=self.tdt, device=self.dev)
        for l, (w, ws) in enumerate(zip(self.pf_layers, self.layers)):
            ops.rownorm(h, w["g1"], None, self.eps, rows=R, C_=H, rms=True, out_act=a, dtype=dt)
            ops.linear(a, w["wqkv"], H, dtype=dt, bias=ws["bqkv"], out_f32=qkv)
            ops.rope_kv_store(qkv, self.inv_freq, pos, q, self.kc[l], self.vc[l], block_table, B=B, rows=Lm,
                              Hq=self.Hq, Hkv=self.Hkv, page=self.page, dtype=dt)
            ops.paged_attn(q, pos, self.kc[l], self.vc[l], block_table, att, B=B, rows=Lm, Hq=self.Hq,
                           Hkv=self.Hkv, page=self.page, dtype=dt)
            ops.linear(att, w["wo"], self.Hq * self.D, dtype=dt, residual=h, out_f32=h)
            ops.rownorm(h, w["g2"], None, self.eps, rows=R, C_=H, rms=True, out_act=a, dtype=dt)
            ops.linear(a, w["wgu"], H, dtype=dt, out_f32=gu)
            ops.swiglu(gu, act, rows=R, I=I, dtype=dt)
            ops.linear(act, w["wdown"], I, dtype=dt, residual=h, out_f32=h)

    def _prefill_chunk(self, xc, pos0, b):
        """<= 64 prompt rows of sequence b through the layers at cache position pos0.  One hipGraph per chunk length over
        static staging buffers: a 50-row prompt is 144 launches, which the host issues in ~1.5 ms eagerly (32 prompts
        in front of a batch: ~50 ms with the GPU mostly idle) and the graph replays in ~0.4 ms."""
        rows = xc.shape[0]
        if self.h2 or self.wplanes:
            # the decode packs are fp16 / weight planes (csrc/decode.hip only): prompt rows go through the windowed GEMM
            hc = xc.to(self.dev, torch.float32).clone()
            self._gemm_layers(hc, 1, rows, torch.tensor([pos0], dtype=torch.int32, device=self.dev), self.block_table[b:b + 1].contiguous())
            return hc, hc.to(self.tdt)
        if not self.use_graphs:
            hc = xc.clone()
            hca = hc.to(self.tdt)
            pos = torch.tensor([pos0], dtype=torch.int32, device=self.dev)
            self._layers(hc, hca, 1, rows, pos, self.block_table[b:b + 1])
            return hc, hca
        if not hasattr(self, "_pf"):
            self._pf = dict(h=torch.zeros(64, self.H, device=self.dev), ha=torch.zeros(64, self.H, dtype=self.tdt, device=self.dev),
                            pos=torch.zeros(1, dtype=torch.int32, device=self.dev),
                            bt=torch.zeros(1, self.max_pages, dtype=torch.int32, device=self.dev), graphs={})
        pf = self._pf
        pf["h"][:rows].copy_(xc)
        pf["ha"][:rows].copy_(xc)
        pf["pos"].fill_(pos0)
        pf["bt"].copy_(self.block_table[b:b + 1])
        if rows not in pf["graphs"]:
            pf["graphs"][rows] = Graphed(lambda r=rows: self._layers(pf["h"][:r], pf["ha"][:r], 1, r, pf["pos"], pf["bt"]), True)
        pf["graphs"][rows]()
        return pf["h"][:rows], pf["ha"][:rows]

    def step(self):
        self._decode()

    def close(self):
        """Deterministic teardown of the recorded graphs (decode step, prompt chunks) on the calling thread."""
        if self._decode is not None:
            self._decode.release()
            self._decode = None
        if hasattr(self, "_pf"):
            for g in self._pf["graphs"].values():
                g.release()
            del self._pf

    @torch.no_grad()
    def forward_rows(self, x: torch.Tensor, pos0: int) -> torch.Tensor:
        """Qwen2Encoder.forward_one_step (llm.py:359-371): appends the rows x [n, H] to sequence 0's KV cache at position
        pos0 (full causal attention over the cache) and returns hidden_states[-1] = the final RMSNorm of
        the backbone, fp32 [n, H]."""
        n = x.shape[0]
        if pos0 + n > self.max_pages * self.page:
            raise RuntimeError("sequence exceeds the KV cache")
        x = x.to(self.dev, torch.float32).contiguous()
        if pos0 == 0:
            self.release(0)
        self._set_pages(0, pos0 + n)
        out = torch.empty(n, self.H, device=self.dev)
        for c0 in range(0, n, 64):
            c1 = min(n, c0 + 64)
            hc, _ = self._prefill_chunk(x[c0:c1], pos0 + c0, 0)
            ops.rownorm(hc, self.norm_w, None, self.eps, rows=c1 - c0, C_=self.H, rms=True, out_f32=out[c0:c1], dtype=F32)
        return out

    def compact_from(self, big: "LlmEngine", idx: List[int]):
        """Continue the still-active sequences `idx` of `big` in this (smaller-batch) engine: loop state, token
        history, next input embedding and block-table rows are gathered into slots 0..len(idx)-1; the KV pages
        themselves are shared and stay where they are.  Unused slots are marked finished."""
        n = len(idx)
        assert n <= self.B and big.kc is self.kc
        ii = torch.tensor(idx, dtype=torch.long, device=self.dev)
        st = torch.zeros(8, self.B, dtype=torch.int32, device=self.dev)
        st[ST_FIN, n:] = 1
        st[:, :n] = big.state[:, ii]
        self.state.copy_(st)
        self.out_tokens[:n].copy_(big.out_tokens[ii])
        self.sampled[:n].copy_(big.sampled[ii])
        self.x_in[:n].copy_(big.x_in[ii])
        self.block_table.fill_(self.trash_page)           # idle slots append their (ignored) KV to the scratch page
        self.block_table[:n].copy_(big.block_table[ii])
        self.seed, self.want_logp = big.seed, False
        self.top_p, self.top_k, self.win_size, self.tau_r = big.top_p, big.top_k, big.win_size, big.tau_r
        self.forced = None
        key = (True, False, self.seed)
        self._fresh_decode_graph(key)

    def _fresh_decode_graph(self, key):
        if self._decode is None or getattr(self, "_graph_key", None) != key:
            if self._decode is not None:
                self._decode.release()
            self._decode = Graphed(self._decode_step, self.use_graphs)
        self._graph_key = key

    def run(self, max_steps: int, poll_every: int = 8) -> List[List[int]]:
        """Decode until every sequence finished or max_steps tokens were tried; returns accepted tokens per sequence."""
        done = 1                                          # start() already sampled step 0
        while done < max_steps:
            n = min(poll_every, max_steps - done)
            if self.reserve_ahead < (1 << 30):
                self.ensure_capacity(n + 1)
            for _ in range(n):
                self._decode()
            done += n
            if bool(self.state[ST_FIN].all().item()):
                break
        return self.tokens()

    @torch.no_grad()
    def run_queue(self, requests, seed=0, poll_every: int = 8, ahead: int = 32):
        """Continuous batching over a queue of requests [(lm_input [L, H], min_len, max_len)]: up to B run at a time; when
        a sequence finishes its pages go back to the allocator and the next queued request is admitted into its slot
        between two decode steps.  Returns the accepted tokens per request (seq id = request index, so the result equals
        running every request alone under the same seed)."""
        self.seed, self.want_logp, self.forced = int(seed), False, None
        self._fresh_decode_graph((True, False, self.seed))
        st = torch.zeros(8, self.B, dtype=torch.int32)
        st[ST_FIN] = 1
        self.state.copy_(st)
        for s_ in range(self.B):
            self.release(s_)
        out: List[Optional[List[int]]] = [None] * len(requests)
        owner = [-1] * self.B
        nxt = 0
        while True:
            fin = self.state[ST_FIN].tolist()
            nout = self.state[ST_NOUT].tolist()
            for s_ in range(self.B):
                if owner[s_] >= 0 and fin[s_]:
                    out[owner[s_]] = self.out_tokens[s_, :nout[s_]].tolist()
                    owner[s_] = -1
                    self.release(s_)
                if owner[s_] < 0 and nxt < len(requests):
                    x, mn, mx = requests[nxt]
                    self.admit(s_, x, mn, mx, seq_id=nxt, ahead=ahead + poll_every)
                    owner[s_] = nxt
                    nxt += 1
            active = [s_ for s_ in range(self.B) if owner[s_] >= 0]
            if not active:
                return out
            self.ensure_capacity(ahead + poll_every, active=active)
            for _ in range(poll_every):
                self._decode()

    # ------------------------------------------------------------------ host-driven stream (bistream decode, llm.py:762-870)
    def open_stream(self, seed=0, seq_id=0, want_logp=False):
        """One sequence (slot 0) whose LM passes are issued one at a time by the host: text rows and speech-token rows
        arrive interleaved, so the loop of llm.py:786-870 stays on the host; every pass still runs on the HIP kernels and
        the sampler stays on the device.  The loop state (fields of include/mmx_hip.h) is mirrored on the host and
        uploaded before each pass."""
        assert self.B == 1, "bistream decode is single-sequence (cli/model.py:105-112)"
        self.release(0)
        self._set_pages(0, self.max_pages * self.page)        # the text arrives incrementally: reserve the whole context
        self.seed, self.want_logp, self.forced = int(seed), bool(want_logp), None
        self._st = dict(rows=0, calls=0, hist=0, seq=int(seq_id), last=None)
        self.sampled.fill_(-1)
        self._fresh_decode_graph((True, self.want_logp, self.seed))

    def embed_text(self, tok: torch.Tensor) -> torch.Tensor:
        """llm.model.model.embed_tokens rows, fp32 [n, H]."""
        tok = tok.reshape(-1).to(self.dev, torch.int64)
        x = torch.empty(tok.numel(), self.H, device=self.dev)
        if tok.numel():
            ops.gather_rows(tok, self.embed_tokens, out_f32=x, dtype=F32)
        return x

    def embed_speech(self, tok: torch.Tensor) -> torch.Tensor:
        tok = tok.reshape(-1).to(self.dev, torch.int64)
        x = torch.empty(tok.numel(), self.H, device=self.dev)
        if tok.numel():
            ops.gather_rows(tok, self.speech_emb, out_f32=x, dtype=F32)
        return x

    def _upload_state(self, pos, ignore_eos):
        s = self._st
        st = torch.tensor([pos, s["calls"], s["hist"], 0, (1 << 30) if ignore_eos else 0, 1 << 30, s["seq"], 0],
                          dtype=torch.int32).reshape(8, 1)
        self.state.copy_(st)

    def feed(self, x: Optional[torch.Tensor], ignore_eos: bool) -> int:
        """One LM pass: appends the rows x [n, H] (None: the embedding of the last accepted token, graph replay) to
        the KV cache and samples from the last row's logits (RAS on the device; ignore_eos as llm.py:259-274).
        Returns the sampled id; call commit() with the id the caller settles on."""
        s = self._st
        n = 1 if x is None else x.shape[0]
        if s["rows"] + n > self.max_pages * self.page or s["calls"] >= self.max_out:
            raise RuntimeError("bistream: sequence exceeds the KV cache (no fill token / eos was produced)")
        if x is None:
            self._upload_state(s["rows"], ignore_eos)
            self._decode()
        else:
            x = x.to(self.dev, torch.float32).contiguous()
            for c0 in range(0, n, 64):
                c1 = min(n, c0 + 64)
                hc, hca = self._prefill_chunk(x[c0:c1], s["rows"] + c0, 0)
            self.h[0].copy_(hc[-1])
            self.h_act[0].copy_(hca[-1])
            self._upload_state(s["rows"] + n - 1, ignore_eos)
            self._tail(1)
        tok = int(self.sampled[0, s["calls"]].item())
        if int(self.state[7, 0].item()):
            raise RuntimeError("sampling reaches max_trials 100 and still get eos when ignore_eos is True, check your input!")
        s["rows"] += n
        s["calls"] += 1
        self._last_sampled = tok
        return tok

    def commit(self, token: int):
        """Appends `token` to the decoded history the repetition window reads (bistream keeps fill / eos ids in it,
        llm.py:831) and makes its embedding the next single-row input when it is a speech id."""
        s = self._st
        if token != self._last_sampled or token >= self.eos:
            self.out_tokens[0, s["hist"]] = token
            if token < self.eos:
                self.x_in[0].copy_(self.speech_emb[token])
            elif self._last_sampled < self.eos and s["last"] is not None:
                self.x_in[0].copy_(self.speech_emb[s["last"]])     # the device draw that was overruled moved x_in
        if token < self.eos:
            s["last"] = token
        s["hist"] += 1

    def tokens(self) -> List[List[int]]:
        n = self.state[ST_NOUT].tolist()
        t = self.out_tokens.cpu()
        return [t[b, :n[b]].tolist() for b in range(self.B)]
