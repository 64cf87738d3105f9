"""CosyVoice2 causal flow-matching decoder engine (FSQ tokens -> DAC-VAE latents) on libmmx_hip kernels.

Reference (speech/): cosyvoice/flow/flow.py:437-511 (inference), transformer/upsample_encoder.py:243-316 (encoder),
flow/flow_matching.py:74-126,323-348 (CFM Euler + CFG), flow/decoder.py:405-496 (estimator),
matcha/models/components/{decoder.py:14-117, transformer.py:243-316}.

Everything is time-major [B, T, C].  fp32 residual streams, compute-dtype GEMM inputs.  Each Linear / causal
Conv1d is one windowed-GEMM launch with bias / activation / residual / mask fused; the estimator's V
projection is computed directly transposed (V^T = W_v X^T, same kernel with operand roles swapped) so the
MFMA flash-attention kernel reads K row-major and V^T row-major without any transpose pass.
A whole 10-step Euler solve (about 5 000 launches) is recorded once per shape into a hipGraph.
"""
import gc
import math
from collections import OrderedDict
from typing import Dict, Optional

import torch

from . import ops
from ._lib import BF16, F32, TORCH_DT, WEIGHT_DT, X2, X2W, is_split


def espnet_rel_pe(T: int, d: int) -> torch.Tensor:
    """embedding.py:233-253 table for relative positions T-1 ... -(T-1): [2T-1, d] fp32 (host, cached)."""
    pos = torch.arange(T - 1, -T, -1, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(2 * T - 1, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


import threading

# Concurrency rule of the engines (one rule, applied everywhere):
#   * every host thread launches on its OWN non-default stream (torch.cuda.stream(...)); the library itself is
#     stream-explicit and keeps no global state (include/mmx_hip.h);
#   * hipGraph capture uses the THREAD-LOCAL capture mode, so launches, graph replays and allocations of other threads
#     proceed while one thread records (they are not captured and do not invalidate the capture);
#   * two captures never overlap: CAPTURE_LOCK is taken by Graphed around the capture itself and by nothing else.
# tests/test_gpu_flow.py::test_capture_while_other_thread_decodes exercises exactly this.
CAPTURE_LOCK = threading.RLock()


_capture_primed = {}


def _prime_capture_state(dev):
    """torch keeps the CUDA generator's capture-state tensors (seed / offset) alive only while at least one CUDAGraph is
    registered with the generator, allocates them at the first registration and updates them IN PLACE at every capture.
    If that allocation happens under torch.inference_mode() (the drop-in modules keep the reference's
    @torch.inference_mode() decorators) they are inference tensors, and a later capture outside inference mode fails
    ("Inplace update to inference tensor outside InferenceMode").  So the first capture of a process is a trivial one
    made with inference mode switched off, and its graph is kept alive for the life of the process: the state tensors
    are then never re-allocated, whatever engines come and go."""
    key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    if key in _capture_primed:
        return
    with torch.inference_mode(False):
        t = torch.zeros(1, device=f"cuda:{key}")
        g = torch.cuda.CUDAGraph()
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()                                   # see Graphed.__call__
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                t.add_(1.0)
        finally:
            if gc_was_on:
                gc.enable()
    _capture_primed[key] = (g, t)


class Graphed:
    """Runs `fn` eagerly once (warm-up), then records it into a hipGraph and replays it.  The capture takes
    CAPTURE_LOCK (captures are serialised against each other only) and uses the thread-local capture mode, so other
    threads keep launching on their own streams meanwhile."""

    cold_calls = 0          # calls (of any instance) that ran eagerly or captured instead of replaying: a caller that times itself
                            # compares the counter before and after (TtsEngine._refit_sched)

    def __init__(self, fn, enabled=True):
        self.fn, self.enabled, self.graph, self.calls = fn, enabled, None, 0

    def release(self):
        """Destroys the recorded graph (and its private pool) NOW, on the calling thread, and drops the closure: the
        owner calls this on eviction / teardown so that no hipGraph is ever left for Python's cyclic collector (which may
        run on a thread that is capturing: see __call__)."""
        g, self.graph, self.fn = self.graph, None, None
        if g is not None:
            with CAPTURE_LOCK:                             # never while another thread is inside a capture
                g.reset()
            del g

    def __call__(self):
        if not self.enabled:
            return self.fn()
        if self.graph is None:
            self.calls += 1
            Graphed.cold_calls += 1
            if self.calls == 1:
                return self.fn()
            with CAPTURE_LOCK:
                torch.cuda.current_stream().synchronize()
                _prime_capture_state(torch.cuda.current_device())
                g = torch.cuda.CUDAGraph()
                # Python's cyclic collector must not run inside the capture: if it finds an unreachable engine there, the
                # destructors of that engine's recorded graphs run on this (capturing) thread, HIP refuses them, the
                # error is thrown from a destructor and the process aborts.  Collect now, switch the collector off for
                # the capture (torch.cuda.graph no longer collects by itself in this torch version).
                gc_was_on = gc.isenabled()
                gc.collect()
                gc.disable()
                try:
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        self.fn()
                finally:
                    if gc_was_on:
                        gc.enable()
                self.graph = g
        self.graph.replay()


class FlowEngine:
    def __init__(self, sd: Dict[str, torch.Tensor], dtype=BF16, device="cuda", n_timesteps=10, cfg_rate=0.7,
                 enc_chunk=25, est_chunk=50, pre_lookahead_len=3, use_graphs=True, parts=("encoder", "estimator"),
                 fused=None, attn="bf16", wplanes=False):
        """wplanes (split build only): every GEMM weight is carried as two bf16 planes hi + lo of the checkpoint's fp32 value
        (MMX_X2W, csrc/gemm.hip) instead of being rounded to bf16 - for checkpoints whose weights are not bf16-representable
        (the reference loads fp32 flow.pt, cli/model.py:67-75).  Three MFMAs per fragment pair (hi*hi + lo*hi + hi*lo) and twice
        the weight stream, in the windowed GEMM and in the fused row-tile kernels alike."""
        self.dtype, self.tdt, self.dev = dtype, TORCH_DT[dtype], torch.device(device)
        self.wplanes = dtype == X2 and ops.resolve_wplanes(wplanes, (v for k, v in sd.items() if v.dim() >= 2 and v.is_floating_point()
                                                                     and k != "input_embedding.weight" and not k.endswith("rand_noise")))
        self.pdt = X2W if self.wplanes else dtype            # the code weights are packed for
        self.n_timesteps, self.cfg, self.L = n_timesteps, cfg_rate, pre_lookahead_len
        self.enc_chunk, self.est_chunk = enc_chunk, est_chunk
        self.use_graphs = use_graphs
        # fused=True: the estimator runs on the row-tile kernels of csrc/fused.hip (2 launches per transformer block,
        # 1 per ResNet block); fused=False: one launch per Linear / Conv1d / LayerNorm (the composition the fused kernels
        # are tested against, tests/test_gpu_kernels.py).  None (default): fused, except for the fp32 build on small
        # problems — exact-fp32 MFMA runs at 1/16 of the bf16 rate, so there the GEMMs are MFMA bound and a row tile per
        # workgroup (64 workgroups for one 10 s utterance) leaves 3/4 of the matrix cores idle, while the per-op launches
        # split N over all CUs (measured, one 10 s utterance, fp32: 175 ms fused, 104 ms per-op)
        self.fused = fused
        self.split = is_split(dtype)
        # attn="fp8": the estimator's full-length attention launches run the fp8 MFMA variant (bf16 build only; BASELINE
        # config 5).  The split-key launches of streaming hops stay bf16.
        assert attn in ("bf16", "fp8")
        self.attn_fp8 = attn == "fp8" and dtype == BF16
        dt = self.pdt
        f = lambda k: sd[k].detach().to(self.dev, torch.float32).contiguous()
        lin = lambda k: ops.pack_linear(f(k), dt)
        cv = lambda k: ops.pack_conv1d(f(k), dt)
        self._pe, self._plans, self._vt = OrderedDict(), OrderedDict(), {}
        self.plan_budget_bytes = 16 << 30           # recorded Euler solves kept alive (LRU); see _cfm_plan
        self.shape_log = None                       # a list collects (n, T, sum T_i, sum T_i^2, polite) of every cfm_batch call (bench.py)
        # polite = True: the fused kernels use 64-row tiles whatever the launch-time model says.  Fewer, longer workgroups:
        # slower for the launch itself below ~8 000 rows, but they occupy fewer CUs, and a latency-bound kernel chain running
        # beside them on another stream (the LM decode loop of TtsEngine.tts_batch) keeps more of the chip
        # (measured, 32-utterance step: decode loop 548 -> 523 ms, step 643 -> 622 ms).  Part of the plan key.
        self.polite = False
        # polite groups of the split build: flash attention on its 128-query workgroups (96 KB of LDS, 136 registers per wave) instead
        # of the 256-query ones (128 KB, 216) the launch-time rule would pick: a decode workgroup fits beside them on the CU
        # (measured in the step: decode loop done at 524 ms against 541, 520.8 against 512.4 audio-s/s; include/mmx_hip.h, form)
        self.polite_flash_form = getattr(FlowEngine, "polite_flash_form_default", 1)
        self.plan_bytes = 0
        # CausalConditionalCFM.__init__: torch CPU manual_seed(0); randn([1,80,15000]) (flow_matching.py:320-321)
        self.rand_noise = torch.randn([1, 80, 50 * 300], generator=torch.Generator().manual_seed(0))
        self.spk_enc = None
        if "speaker_encoder.init.weight" in sd:
            from .spk import SpeakerEncoderEngine
            self.spk_enc = SpeakerEncoderEngine(sd, dtype=dtype, device=device, pack_dtype=self.pdt)
        if "encoder" in parts:
            self._init_encoder(sd, f, lin, cv)
        if "estimator" in parts:
            self._init_estimator(sd, f, lin, cv)

    def clone_shared(self):
        """A second engine over the SAME packed weights with its own plan / scratch buffers, for a second host
        thread + stream (plans hold the static buffers of a recorded graph, so they cannot be shared)."""
        import copy
        c = copy.copy(self)
        c._plans, c._pe, c._vt, c.plan_bytes = OrderedDict(), OrderedDict(), {}, 0
        c._stream_pool = OrderedDict()
        return c

    def set_noise(self, noise: torch.Tensor):
        """Replaces rand_noise (the drop-in CausalConditionalCFM owns its own tensor, flow_matching.py:321)."""
        if noise is not self.rand_noise and not torch.equal(noise.cpu(), self.rand_noise):
            self.rand_noise = noise.detach().cpu().float()
            self.close()                                   # recorded plans baked the old noise in

    def _init_encoder(self, sd, f, lin, cv):
        dt = self.dtype
        flow_level = "input_embedding.weight" in sd          # absent when built for a bare UpsampleConformerEncoder
        if flow_level:
            self.emb_table = f("input_embedding.weight")
            self.spk_w, self.spk_b = lin("spk_embed_affine_layer.weight"), f("spk_embed_affine_layer.bias")
            self.spk_dim = sd["spk_embed_affine_layer.weight"].shape[1]
            self.spk_gamma = torch.full((self.spk_dim,), 1.0 / math.sqrt(self.spk_dim), device=self.dev)
        e = "encoder"
        xs = math.sqrt(512.0)

        def embed(p):
            return dict(w=lin(p + ".out.0.weight"), b=f(p + ".out.0.bias"), g=f(p + ".out.1.weight") * xs,
                        beta=f(p + ".out.1.bias") * xs)

        def conf_layer(p):
            a = p + ".self_attn"
            return dict(
                n1g=f(p + ".norm_mha.weight"), n1b=f(p + ".norm_mha.bias"),
                wqkv=ops.pack_linear(torch.cat([f(a + ".linear_q.weight"), f(a + ".linear_k.weight"), f(a + ".linear_v.weight")], 0), self.pdt),
                bqkv=torch.cat([f(a + ".linear_q.bias"), f(a + ".linear_k.bias"), f(a + ".linear_v.bias")], 0).contiguous(),
                wpos=lin(a + ".linear_pos.weight"), pu=f(a + ".pos_bias_u"), pv=f(a + ".pos_bias_v"),
                wo=lin(a + ".linear_out.weight"), bo=f(a + ".linear_out.bias"),
                n2g=f(p + ".norm_ff.weight"), n2b=f(p + ".norm_ff.bias"),
                w1=lin(p + ".feed_forward.w_1.weight"), b1=f(p + ".feed_forward.w_1.bias"),
                w2=lin(p + ".feed_forward.w_2.weight"), b2=f(p + ".feed_forward.w_2.bias"))

        # bf16 build: rel-pos attention on the MFMA (mmx_attn_relpos_bf16) - Q | K rows from one GEMM, V^T from a second one
        # with the weights as its A operand (the layout the flash kernels read); other builds: mmx_attn_dense (fp32 VALU)
        self.enc_mfma = dt == BF16 and getattr(self, "enc_attn", "mfma") == "mfma"
        self.enc = dict(embed=embed(e + ".embed"), up_embed=embed(e + ".up_embed"),
                        pl_w1=cv(e + ".pre_lookahead_layer.conv1.weight"), pl_b1=f(e + ".pre_lookahead_layer.conv1.bias"),
                        pl_w2=cv(e + ".pre_lookahead_layer.conv2.weight"), pl_b2=f(e + ".pre_lookahead_layer.conv2.bias"),
                        layers=[conf_layer(f"{e}.encoders.{i}") for i in range(6)],
                        up_w=cv(e + ".up_layer.conv.weight"), up_b=f(e + ".up_layer.conv.bias"),
                        up_layers=[conf_layer(f"{e}.up_encoders.{i}") for i in range(4)],
                        ang=f(e + ".after_norm.weight"), anb=f(e + ".after_norm.bias"))
        if flow_level:
            self.enc.update(wproj=lin("encoder_proj.weight"), bproj=f("encoder_proj.bias"))

    def _init_estimator(self, sd, f, lin, cv):
        dt = self.dtype
        q = "decoder.estimator"
        self.tdim = sd[q + ".time_mlp.linear_1.weight"].shape[1]             # 320
        self.t_w1, self.t_b1 = lin(q + ".time_mlp.linear_1.weight"), f(q + ".time_mlp.linear_1.bias")
        self.t_w2, self.t_b2 = lin(q + ".time_mlp.linear_2.weight"), f(q + ".time_mlp.linear_2.bias")

        def tblock(p):
            a = p + ".attn1"
            d = dict(n1g=f(p + ".norm1.weight"), n1b=f(p + ".norm1.bias"),
                     wo=lin(a + ".to_out.0.weight"), bo=f(a + ".to_out.0.bias"),
                     n3g=f(p + ".norm3.weight"), n3b=f(p + ".norm3.bias"),
                     w1=lin(p + ".ff.net.0.proj.weight"), b1=f(p + ".ff.net.0.proj.bias"),
                     w2=lin(p + ".ff.net.2.weight"), b2=f(p + ".ff.net.2.bias"))
            wq, wk, wv = f(a + ".to_q.weight"), f(a + ".to_k.weight"), f(a + ".to_v.weight")
            if self.fused is not False:
                pk = (lambda w: ops.pack_skinny(w.float().contiguous(), dtype=X2W)) if self.wplanes else \
                     (lambda w: ops.pack_skinny(w.to(WEIGHT_DT[dt]).contiguous(), dtype=dt))
                d.update(wo_p=pk(f(a + ".to_out.0.weight")), w1_p=pk(f(p + ".ff.net.0.proj.weight")),
                         w2_p=pk(f(p + ".ff.net.2.weight")), wqkv_p=pk(torch.cat([wq, wk, wv], 0)))
            if dt == BF16:
                d["wqk"] = ops.pack_linear(torch.cat([wq, wk], 0), dt)       # [1024, 256]
                d["wv"] = ops.pack_linear(wv, dt)                            # A operand of the V^T GEMM
            else:
                d["wqkv"] = ops.pack_linear(torch.cat([wq, wk, wv], 0), self.pdt)
            return d

        self.resnets, mlp_w, mlp_b = [], [], []

        def resnet(p):
            idx = len(self.resnets)
            r = dict(idx=idx, cin=sd[p + ".block1.block.0.weight"].shape[1],
                     w1=cv(p + ".block1.block.0.weight"), b1=f(p + ".block1.block.0.bias"),
                     g1=f(p + ".block1.block.2.weight"), be1=f(p + ".block1.block.2.bias"),
                     w2=cv(p + ".block2.block.0.weight"), b2=f(p + ".block2.block.0.bias"),
                     g2=f(p + ".block2.block.2.weight"), be2=f(p + ".block2.block.2.bias"),
                     wr=cv(p + ".res_conv.weight"), br=f(p + ".res_conv.bias"))
            if self.fused is not False:
                pc = (lambda k: ops.pack_skinny(ops.conv1d_matrix(f(k)).contiguous(), dtype=X2W)) if self.wplanes else \
                     (lambda k: ops.pack_skinny(ops.pack_conv1d(f(k), dt), dtype=dt))
                r.update(w1_p=pc(p + ".block1.block.0.weight"), w2_p=pc(p + ".block2.block.0.weight"), wr_p=pc(p + ".res_conv.weight"))
            mlp_w.append(f(p + ".mlp.1.weight"))
            mlp_b.append(f(p + ".mlp.1.bias"))
            self.resnets.append(r)
            return r

        def stage(p):
            return dict(res=resnet(p + ".0"), blocks=[tblock(f"{p}.1.{j}") for j in range(4)])

        self.n_mid = len({k.split(".")[3] for k in sd if k.startswith(q + ".mid_blocks.")})
        self.down = stage(q + ".down_blocks.0")
        self.down_w, self.down_b = cv(q + ".down_blocks.0.2.weight"), f(q + ".down_blocks.0.2.bias")
        self.mid = [stage(f"{q}.mid_blocks.{i}") for i in range(self.n_mid)]
        self.up = stage(q + ".up_blocks.0")
        self.up_w, self.up_b = cv(q + ".up_blocks.0.2.weight"), f(q + ".up_blocks.0.2.bias")
        self.fin_w, self.fin_b = cv(q + ".final_block.block.0.weight"), f(q + ".final_block.block.0.bias")
        self.fin_g, self.fin_be = f(q + ".final_block.block.2.weight"), f(q + ".final_block.block.2.bias")
        self.proj_w, self.proj_b = cv(q + ".final_proj.weight"), f(q + ".final_proj.bias")
        self.mlp_w = ops.pack_linear(torch.cat(mlp_w, 0), self.pdt)          # [14*256, 1024]
        self.mlp_b = torch.cat(mlp_b, 0).contiguous()
        self.C = sd[q + ".final_proj.weight"].shape[1]

    # ------------------------------------------------------------------ helpers
    def _new(self, *shape, f32=False):
        return torch.empty(*shape, dtype=torch.float32 if f32 else self.tdt, device=self.dev)

    def _pos(self, T):
        """rel-pos table of length T on the device; a small LRU (a streaming utterance asks for a new T at every hop)."""
        if T in self._pe:
            self._pe.move_to_end(T)
        else:
            self._pe[T] = espnet_rel_pe(T, 512).to(self.dev, self.tdt)
            while len(self._pe) > 64:
                self._pe.popitem(last=False)
        return self._pe[T]

    # ------------------------------------------------------------------ encoder
    def _conformer(self, lw, x, T, pos_act, chunk, B=1, klen=None, keymask=None):
        """One conformer layer on x fp32 [B * T, 512] (B utterances zero padded to T rows each; klen int32 [B] / keymask fp32
        [B, T]: the valid rows - the rows beyond them carry finite garbage that no valid row ever reads)."""
        dt = self.dtype
        R = B * T
        hn = self._new(R, 512)
        ops.rownorm(x, lw["n1g"], lw["n1b"], 1e-12, rows=R, C_=512, out_act=hn, dtype=dt)
        p = self._new(2 * T - 1, 512)
        ops.linear(pos_act, lw["wpos"], 512, dtype=dt, out_act=p)
        ao = self._new(R, 512)
        if self.enc_mfma:
            qk = self._new(R, 1024)
            ops.linear(hn, lw["wqkv"][:1024], 512, dtype=dt, bias=lw["bqkv"][:1024], out_act=qk)
            Tp = ops.round_up(T, 8)
            vt = self._new(B, 512, Tp)                        # [B][512][Tp]; not from the plan-lifetime cache (_vt_buf):
            if Tp > T:                                        # an encoder call's shape is arbitrary, the cache would only grow
                vt[:, :, T:].zero_()                          # pad columns must be finite: the flash tiles read 8 at a time
            ops.gemm(lw["wqkv"][1024:], hn, 512, T, dtype=dt, lda=lw["wqkv"].shape[1], cin=512, batch=B, a_bstride=0,
                     w_bstride=T * 512, bias=lw["bqkv"][1024:], bias_per_row=True, out_act=vt, ldo_a=Tp, oa_bstride=512 * Tp)
            ops.attn_relpos_bf16(qk, qk[:, 512:], vt, p, lw["pu"], lw["pv"], ao, B=B, H=8, T=T, ldq=1024, ldk=1024, ldvt=Tp,
                                 ldp=512, ldo=512, q_bs=T * 1024, k_bs=T * 1024, vt_bs=512 * Tp, o_bs=T * 512, scale=0.125,
                                 chunk=chunk, klen=klen)
        elif self.split and getattr(self, "enc_attn", "mfma") == "mfma" and (klen is not None or keymask is None):
            # split build: the same attention on the MFMA with every operand as bf16 hi + lo (mmx_attn_relpos_x; prefix masks
            # only: the batched encoder's klen)
            qkv = self._new(R, 1536)
            ops.linear(hn, lw["wqkv"], 512, dtype=dt, bias=lw["bqkv"], out_act=qkv)
            ops.attn_relpos_x(qkv, qkv[:, 512:], qkv[:, 1024:], p, lw["pu"], lw["pv"], ao, B=B, H=8, T=T, ldq=1536, ldk=1536, ldv=1536,
                              ldp=512, ldo=512, q_bs=T * 1536, k_bs=T * 1536, v_bs=T * 1536, o_bs=T * 512, scale=0.125, chunk=chunk, klen=klen)
        else:
            qkv = self._new(R, 1536)
            ops.linear(hn, lw["wqkv"], 512, dtype=dt, bias=lw["bqkv"], out_act=qkv)
            ops.attn_dense(qkv, qkv[:, 512:], qkv[:, 1024:], ao, B=B, H=8, Tq=T, Tk=T, ldq=1536, ldk=1536, ldv=1536, ldo=512,
                           q_bs=T * 1536, k_bs=T * 1536, v_bs=T * 1536, o_bs=T * 512, scale=0.125, dtype=dt, chunk=chunk, pos=p,
                           ldp=512, pos_u=lw["pu"], pos_v=lw["pv"], keymask=keymask)
        x2 = self._new(R, 512, f32=True)
        ops.linear(ao, lw["wo"], 512, dtype=dt, bias=lw["bo"], residual=x, out_f32=x2)
        ops.rownorm(x2, lw["n2g"], lw["n2b"], 1e-12, rows=R, C_=512, out_act=hn, dtype=dt)
        ff = self._new(R, 2048)
        ops.linear(hn, lw["w1"], 512, dtype=dt, bias=lw["b1"], act="silu", out_act=ff)
        x3 = self._new(R, 512, f32=True)
        ops.linear(ff, lw["w2"], 2048, dtype=dt, bias=lw["b2"], residual=x2, out_f32=x3)
        return x3

    def _embed(self, ew, a, T, rowmask=None):
        """LinearNoSubsampling + LayerNorm (eps 1e-5) * sqrt(512): act [T,512] -> fp32 [T,512] and act copy (* rowmask)."""
        dt = self.dtype
        tmp = self._new(T, 512, f32=True)
        ops.linear(a, ew["w"], 512, dtype=dt, bias=ew["b"], out_f32=tmp)
        x = self._new(T, 512, f32=True)
        xa = self._new(T, 512)
        ops.rownorm(tmp, ew["g"], ew["beta"], 1e-5, rows=T, C_=512, out_f32=x, out_act=xa, dtype=dt, rowmask=rowmask)
        return x, xa

    def encode(self, ids: torch.Tensor, finalize: bool, streaming: bool) -> torch.Tensor:
        """ids [L] int64 (prompt + tokens). Returns mu fp32 time-major [2*T, 80], T = L (finalize) or L - 3."""
        Lt = ids.numel()
        a0 = self._new(Lt, 512)
        ops.gather_rows(ids, self.emb_table, out_act=a0, dtype=self.dtype)
        return self.encode_embedded(a0, finalize, streaming)

    def encode_embedded(self, a0: torch.Tensor, finalize: bool, streaming: bool, hidden=False) -> torch.Tensor:
        """UpsampleConformerEncoder.forward (upsample_encoder.py:243-316) on embedded rows a0 [Lt, 512] (compute dtype);
        with finalize=False the last 3 rows are the look-ahead `context`.  Returns mu = encoder_proj(h) fp32 [2T, 80],
        or with hidden=True the encoder output h itself (after after_norm) as fp32 [2T, 512]."""
        dt, E = self.dtype, self.enc
        Lt = a0.shape[0]
        x_all, xa_all = self._embed(E["embed"], a0, Lt)
        T = Lt if finalize else Lt - self.L
        rows_in = Lt                                    # look-ahead context rows follow the T rows contiguously
        # PreLookaheadLayer: conv k4 over [x ; context|zeros] -> leaky_relu(0.01) -> causal conv k3 -> + x
        h1 = self._new(T, 512)
        ops.gemm(xa_all, E["pl_w1"], T, 512, dtype=dt, lda=512, cin=512, ntaps=4, row_off=0, row_lo=0, row_hi=rows_in,
                 bias=E["pl_b1"], act="lrelu", slope=0.01, out_act=h1, ldo_a=512)
        x = self._new(T, 512, f32=True)
        ops.conv1d(h1, E["pl_w2"], T=T, Cin=512, k=3, pad_left=2, dtype=dt, bias=E["pl_b2"], residual=x_all, out_f32=x)
        pos = self._pos(T)
        chunk = self.enc_chunk if streaming else 0
        for lw in E["layers"]:
            x = self._conformer(lw, x, T, pos, chunk)
        # Upsample1D: nearest x2, left pad 4, conv k5
        T2 = 2 * T
        up = self._new(T2, 512)
        ops.copy2d(x, F32, 0, 512, 1, up, dt, 0, 512, 1, rows=T2, cols=512, rep=2)
        c5 = self._new(T2, 512)
        ops.conv1d(up, E["up_w"], T=T2, Cin=512, k=5, pad_left=4, dtype=dt, bias=E["up_b"], out_act=c5)
        x, _ = self._embed(E["up_embed"], c5, T2)
        pos = self._pos(T2)
        for lw in E["up_layers"]:
            x = self._conformer(lw, x, T2, pos, 2 * chunk)
        hn = self._new(T2, 512)
        hf = self._new(T2, 512, f32=True) if hidden else None
        ops.rownorm(x, E["ang"], E["anb"], 1e-5, rows=T2, C_=512, out_f32=hf, out_act=hn, dtype=dt)
        if hidden:
            return hf
        mu = self._new(T2, 80, f32=True)
        ops.linear(hn, E["wproj"], 512, dtype=dt, bias=E["bproj"], out_f32=mu)
        return mu

    def encode_batch(self, ids_list) -> list:
        """UpsampleConformerEncoder.forward + encoder_proj for several whole utterances at once (finalize, no chunk masks):
        the token rows are zero padded to the longest, every launch covers the batch (the encoder is ~115 launches, most of
        them too small to fill the chip one utterance at a time).  What makes the padding invisible to the valid rows:
        the embedding output is masked to zero beyond each length (the look-ahead conv reads 3 rows to the right - zeros, as
        at the end of a lone utterance), the other convs only look left, attention masks the keys beyond the length, and the
        relative position of a (query, key) pair does not depend on the sequence length.  Returns mu fp32 [2 * L_b, 80] per
        utterance (views of one buffer)."""
        dt, E = self.dtype, self.enc
        B = len(ids_list)
        lens = [int(i.numel()) for i in ids_list]
        if B == 1:
            return [self.encode(ids_list[0].reshape(-1), True, False)]
        T = max(lens)
        ids = torch.zeros(B, T, dtype=torch.int64, device=self.dev)
        for b, i in enumerate(ids_list):
            ids[b, :lens[b]] = i.reshape(-1)
        lens_t = torch.tensor(lens, dtype=torch.int32, device=self.dev)
        ar = torch.arange(2 * T, device=self.dev)
        mask = (ar[None, :T] < lens_t[:, None]).float().contiguous()             # [B, T]
        mask2 = (ar[None, :] < 2 * lens_t[:, None]).float().contiguous()         # [B, 2T]
        klen2 = (2 * lens_t).contiguous()
        R = B * T
        a0 = self._new(R, 512)
        ops.gather_rows(ids.reshape(-1), self.emb_table, out_act=a0, dtype=dt)
        x_all, xa_all = self._embed(E["embed"], a0, R, rowmask=mask.reshape(-1))
        h1 = self._new(R, 512)
        ops.gemm(xa_all, E["pl_w1"], T, 512, dtype=dt, lda=512, cin=512, ntaps=4, row_off=0, row_lo=0, row_hi=T, batch=B,
                 a_bstride=T * 512, bias=E["pl_b1"], act="lrelu", slope=0.01, out_act=h1, ldo_a=512, oa_bstride=T * 512)
        x = self._new(R, 512, f32=True)
        ops.conv1d(h1, E["pl_w2"], T=T, Cin=512, k=3, pad_left=2, dtype=dt, batch=B, bias=E["pl_b2"], residual=x_all, out_f32=x)
        pos = self._pos(T)
        km, km2 = (None, None) if self.enc_mfma else (mask, mask2)
        for lw in E["layers"]:
            x = self._conformer(lw, x, T, pos, 0, B=B, klen=lens_t, keymask=km)
        T2 = 2 * T
        up = self._new(B * T2, 512)
        ops.copy2d(x, F32, T * 512, 512, 1, up, dt, T2 * 512, 512, 1, rows=T2, cols=512, batch=B, rep=2)
        c5 = self._new(B * T2, 512)
        ops.conv1d(up, E["up_w"], T=T2, Cin=512, k=5, pad_left=4, dtype=dt, batch=B, bias=E["up_b"], out_act=c5)
        x, _ = self._embed(E["up_embed"], c5, B * T2)
        pos = self._pos(T2)
        for lw in E["up_layers"]:
            x = self._conformer(lw, x, T2, pos, 0, B=B, klen=klen2, keymask=km2)
        hn = self._new(B * T2, 512)
        ops.rownorm(x, E["ang"], E["anb"], 1e-5, rows=B * T2, C_=512, out_act=hn, dtype=dt)
        mu = self._new(B, T2, 80, f32=True)
        ops.linear(hn, E["wproj"], 512, dtype=dt, bias=E["bproj"], out_f32=mu)
        return [mu[b, :2 * lens[b]] for b in range(B)]

    # ------------------------------------------------------------------ streaming encoder with cached state
    def _enc_stream_state(self, st):
        """The conformer encoder is chunk-causal in streaming mode (25-token / 50-frame chunks, 3 tokens of look-ahead
        supplied as context), so a hop only has to encode its new tokens: kept per utterance are every layer's Q|K|V
        rows, the inputs of the three convs (look-ahead k4, causal k3, upsample k5) and the projected relative-position
        tables (computed once for the whole capacity: row Tcap - 1 - r is relative distance r for every length)."""
        if st.enc is None:
            E, dt = self.enc, self.dtype
            Tt, Tf = st.Tcap // 2 + self.L, st.Tcap          # token / frame capacity
            z = lambda *sh, f32=False: torch.zeros(*sh, dtype=torch.float32 if f32 else self.tdt, device=self.dev)

            def pos_tables(layers, T):
                pe = espnet_rel_pe(T, 512).to(self.dev, self.tdt)
                out = []
                for lw in layers:
                    p = z(2 * T - 1, 512)
                    ops.linear(pe, lw["wpos"], 512, dtype=dt, out_act=p)
                    out.append(p)
                return out

            st.enc = dict(T=0, xa=z(Tt, 512), x=z(Tt, 512, f32=True), h1=z(Tt, 512), up=z(Tf, 512), mu=z(Tf, 80, f32=True),
                          qkv=[z(Tt, 1536) for _ in E["layers"]], qkv2=[z(Tf, 1536) for _ in E["up_layers"]],
                          pos=pos_tables(E["layers"], Tt), pos2=pos_tables(E["up_layers"], Tf), Tt=Tt, Tf=Tf)
        return st.enc

    def _conformer_stream(self, lw, x, t0, T, qkv, pos_all, Tcap, chunk):
        """One ConformerEncoderLayer on rows t0 .. T-1 (x fp32 [T - t0, 512] window, returned likewise); qkv [Tcap, 1536]
        holds the Q|K|V rows of ALL positions, pos_all [2*Tcap - 1, 512] the projected rel-pos table of the capacity."""
        dt, n = self.dtype, T - t0
        hn = self._new(n, 512)
        ops.rownorm(x, lw["n1g"], lw["n1b"], 1e-12, rows=n, C_=512, out_act=hn, dtype=dt)
        ops.linear(hn, lw["wqkv"], 512, dtype=dt, bias=lw["bqkv"], out_act=qkv[t0:T])
        ao = self._new(T, 512)
        # the kernel indexes pos[T - 1 - i + j]: shift the base so that this is row Tcap - 1 - (i - j) of the full table
        ops.attn_dense(qkv, qkv[:, 512:], qkv[:, 1024:], ao, B=1, H=8, Tq=T, Tk=T, ldq=1536, ldk=1536, ldv=1536, ldo=512,
                       q_bs=0, k_bs=0, v_bs=0, o_bs=0, scale=0.125, dtype=dt, chunk=chunk, pos=pos_all[Tcap - T:], ldp=512,
                       pos_u=lw["pu"], pos_v=lw["pv"], q_begin=t0)
        x2 = self._new(n, 512, f32=True)
        ops.linear(ao[t0:T], lw["wo"], 512, dtype=dt, bias=lw["bo"], residual=x, out_f32=x2)
        ops.rownorm(x2, lw["n2g"], lw["n2b"], 1e-12, rows=n, C_=512, out_act=hn, dtype=dt)
        ff = self._new(n, 2048)
        ops.linear(hn, lw["w1"], 512, dtype=dt, bias=lw["b1"], act="silu", out_act=ff)
        x3 = self._new(n, 512, f32=True)
        ops.linear(ff, lw["w2"], 2048, dtype=dt, bias=lw["b2"], residual=x2, out_f32=x3)
        return x3

    def _encode_stream_rows(self, st, ids: torch.Tensor, tb: int, Lt: int):
        """Streaming, non-final encode of ids [Lt] (the last 3 are look-ahead context) given that tokens 0 .. tb-1 are
        already encoded in the state: only tokens tb .. Lt-4 run through the layers; mu of their frames lands in the
        state's mu buffer.  Pure device work on state buffers (recorded into the per-hop hipGraph)."""
        dt, E, S = self.dtype, self.enc, self._enc_stream_state(st)
        T = Lt - self.L
        assert tb < T and Lt <= S["Tt"]
        # embed the new tokens and the context rows (the previous hop's context rows are re-embedded: they are tokens now)
        a0 = self._new(Lt - tb, 512)
        ops.gather_rows(ids[tb:], self.emb_table, out_act=a0, dtype=dt)
        tmp = self._new(Lt - tb, 512, f32=True)
        ops.linear(a0, E["embed"]["w"], 512, dtype=dt, bias=E["embed"]["b"], out_f32=tmp)
        ops.rownorm(tmp, E["embed"]["g"], E["embed"]["beta"], 1e-5, rows=Lt - tb, C_=512, out_f32=S["x"][tb:Lt], out_act=S["xa"][tb:Lt], dtype=dt)
        # PreLookaheadLayer on rows tb .. T-1: conv k4 over [x ; context], leaky_relu, causal conv k3, + x
        n = T - tb
        ops.gemm(S["xa"], E["pl_w1"], n, 512, dtype=dt, lda=512, cin=512, ntaps=4, row_off=tb, row_lo=0, row_hi=Lt,
                 bias=E["pl_b1"], act="lrelu", slope=0.01, out_act=S["h1"][tb:T], ldo_a=512)
        x = self._new(n, 512, f32=True)
        ops.gemm(S["h1"], E["pl_w2"], n, 512, dtype=dt, lda=512, cin=512, ntaps=3, row_off=tb - 2, row_lo=0, row_hi=T,
                 bias=E["pl_b2"], residual=S["x"][tb:T], ldr=512, out_f32=x, ldo_f=512)
        for lw, qkv, pos in zip(E["layers"], S["qkv"], S["pos"]):
            x = self._conformer_stream(lw, x, tb, T, qkv, pos, S["Tt"], self.enc_chunk)
        # Upsample1D: nearest x2, causal conv k5 (left pad 4) on frames 2tb .. 2T-1
        f0, F = 2 * tb, 2 * T
        ops.copy2d(x, F32, 0, 512, 1, S["up"][f0:F], dt, 0, 512, 1, rows=F - f0, cols=512, rep=2)
        c5 = self._new(F - f0, 512)
        ops.gemm(S["up"], E["up_w"], F - f0, 512, dtype=dt, lda=512, cin=512, ntaps=5, row_off=f0 - 4, row_lo=0, row_hi=F,
                 bias=E["up_b"], out_act=c5, ldo_a=512)
        x, _ = self._embed(E["up_embed"], c5, F - f0)
        for lw, qkv, pos in zip(E["up_layers"], S["qkv2"], S["pos2"]):
            x = self._conformer_stream(lw, x, f0, F, qkv, pos, S["Tf"], 2 * self.enc_chunk)
        hn = self._new(F - f0, 512)
        ops.rownorm(x, E["ang"], E["anb"], 1e-5, rows=F - f0, C_=512, out_act=hn, dtype=dt)
        ops.linear(hn, E["wproj"], 512, dtype=dt, bias=E["bproj"], out_f32=S["mu"][f0:F])

    # ------------------------------------------------------------------ estimator
    def _resnet(self, r, a_in, lda, B, T, tv, mask, out_x):
        """CausalResnetBlock1D; a_in: act [B,T,lda] holding x*mask; writes fp32 out_x [B,T,256]."""
        dt, C = self.dtype, self.C
        cin = r["cin"]
        c1 = self._new(B, T, C, f32=True)
        kw = dict(dtype=dt, lda=lda, cin=cin, ntaps=3, row_off=-2, row_lo=0, row_hi=T, batch=B, a_bstride=T * lda)
        ops.gemm(a_in, r["w1"], T, C, bias=r["b1"], out_f32=c1, ldo_f=C, of_bstride=T * C, **kw)
        h1 = self._new(B, T, C)
        ops.rownorm(c1, r["g1"], r["be1"], 1e-5, rows=T, C_=C, batch=B, act="mish", rowmask=mask,
                    addvec=tv[:, r["idx"] * C:], av_bstride=tv.shape[1], out_act=h1, dtype=dt)
        ops.conv1d(h1, r["w2"], T=T, Cin=C, k=3, pad_left=2, dtype=dt, batch=B, bias=r["b2"], out_f32=c1)
        h2 = self._new(B, T, C, f32=True)
        ops.rownorm(c1, r["g2"], r["be2"], 1e-5, rows=T, C_=C, batch=B, act="mish", rowmask=mask, out_f32=h2, dtype=dt)
        ops.gemm(a_in, r["wr"], T, C, dtype=dt, lda=lda, cin=cin, ntaps=1, row_lo=0, row_hi=T, batch=B, a_bstride=T * lda,
                 bias=r["br"], residual=h2, ldr=C, r_bstride=T * C, out_f32=out_x, ldo_f=C, of_bstride=T * C)

    def _tblock(self, w, x, B, T, mask, chunk, act_out=None, act_ld=0, klen=None):
        """BasicTransformerBlock on fp32 x [B,T,256] (in place); optional compute-dtype copy of the result
        (x*mask) into act_out with row stride act_ld."""
        dt, C = self.dtype, self.C
        hn = self._new(B, T, C)
        ops.rownorm(x, w["n1g"], w["n1b"], 1e-5, rows=T, C_=C, batch=B, out_act=hn, dtype=dt)
        ao = self._new(B, T, 512)
        if dt == BF16:
            qk = self._new(B, T, 1024)
            ops.gemm(hn, w["wqk"], T, 1024, dtype=dt, lda=C, cin=C, batch=B, a_bstride=T * C, out_act=qk, ldo_a=1024,
                     oa_bstride=T * 1024)
            Tp = ops.round_up(T, 8)
            vt = self._vt_buf(B, Tp)
            # V^T[b] = W_v (512x256) . X_b^T : A = weights, "W" operand = activations (batch stride on W)
            ops.gemm(w["wv"], hn, 512, T, dtype=dt, lda=w["wv"].shape[1], cin=C, batch=B, a_bstride=0, w_bstride=T * C,
                     out_act=vt, ldo_a=Tp, oa_bstride=512 * Tp)
            ops.attn_flash_bf16(qk, qk[:, :, 512:], vt, ao, B=B, H=8, T=T, ldq=1024, ldk=1024, ldvt=Tp, ldo=512,
                                q_bs=T * 1024, k_bs=T * 1024, vt_bs=512 * Tp, o_bs=T * 512, scale=0.125, keymask=mask,
                                chunk=chunk)
        else:
            qkv = self._new(B, T, 1536)
            ops.gemm(hn, w["wqkv"], T, 1536, dtype=dt, lda=C, cin=C, batch=B, a_bstride=T * C, out_act=qkv, ldo_a=1536,
                     oa_bstride=T * 1536)
            if self.split:
                ops.attn_flash_x(qkv, qkv[:, :, 512:], qkv[:, :, 1024:], ao, B=B, H=8, T=T, ldq=1536, ldk=1536, ldv=1536, ldo=512,
                                 q_bs=T * 1536, k_bs=T * 1536, v_bs=T * 1536, o_bs=T * 512, scale=0.125,
                                 keymask=(None if klen is not None else mask), chunk=chunk, klen=klen)
            else:
                ops.attn_dense(qkv, qkv[:, :, 512:], qkv[:, :, 1024:], ao, B=B, H=8, Tq=T, Tk=T, ldq=1536, ldk=1536, ldv=1536,
                               ldo=512, q_bs=T * 1536, k_bs=T * 1536, v_bs=T * 1536, o_bs=T * 512, scale=0.125, dtype=dt,
                               keymask=mask, chunk=chunk)
        ops.gemm(ao, w["wo"], T, C, dtype=dt, lda=512, cin=512, batch=B, a_bstride=T * 512, bias=w["bo"], residual=x,
                 ldr=C, r_bstride=T * C, out_f32=x, ldo_f=C, of_bstride=T * C)
        ops.rownorm(x, w["n3g"], w["n3b"], 1e-5, rows=T, C_=C, batch=B, out_act=hn, dtype=dt)
        ff = self._new(B, T, 1024)
        ops.gemm(hn, w["w1"], T, 1024, dtype=dt, lda=C, cin=C, batch=B, a_bstride=T * C, bias=w["b1"], act="gelu",
                 out_act=ff, ldo_a=1024, oa_bstride=T * 1024)
        ops.gemm(ff, w["w2"], T, C, dtype=dt, lda=1024, cin=1024, batch=B, a_bstride=T * 1024, bias=w["b2"], residual=x,
                 ldr=C, r_bstride=T * C, out_f32=x, ldo_f=C, of_bstride=T * C,
                 rowmask=(mask if act_out is not None else None), rm_bstride=T,
                 out_act=act_out, ldo_a=act_ld, oa_bstride=T * act_ld)

    def _vt_buf(self, B, Tp, planes=1):
        key = (B, Tp, planes)                          # referenced by recorded graphs: never evicted (<= 15 MB each)
        if key not in self._vt:
            self._vt[key] = torch.zeros(B, planes * 512, Tp, dtype=torch.bfloat16 if planes > 1 else self.tdt, device=self.dev)   # pad columns stay zero
        return self._vt[key]

    def estimator(self, x, x_bstride, mu, spks, cond, t, B, T, mask=None, streaming=False, out=None, x_mod=None, klen=None):
        """All inputs fp32 time-major device tensors: x [x_mod,T,80] (batch b reads x[b % x_mod]), mu/cond [B,T,80],
        spks [B,80], t [B]; mask fp32 [B,T] or None.  Returns fp32 [B,T,80]."""
        if self.fused or (self.fused is None and (self.dtype == BF16 or self.split or B * ((T + 15) // 16) >= 256)):
            return self._estimator_fused(x, x_bstride, mu, spks, cond, t, B, T, mask, streaming, out, x_mod, klen)
        dt, C = self.dtype, self.C
        chunk = self.est_chunk if streaming else 0
        te = self._new(B, self.tdim)
        ops.sinusoidal_emb(t, te, dim=self.tdim, dtype=dt)
        t1 = self._new(B, 1024)
        ops.linear(te, self.t_w1, self.tdim, dtype=dt, bias=self.t_b1, act="silu", out_act=t1)
        t2 = self._new(B, 1024)
        ops.linear(t1, self.t_w2, 1024, dtype=dt, bias=self.t_b2, act2="mish", out_act=t2)       # mish(time_mlp(t))
        tv = self._new(B, self.mlp_w.shape[0], f32=True)
        ops.linear(t2, self.mlp_w, 1024, dtype=dt, bias=self.mlp_b, out_f32=tv)                  # all 14 resnet mlps
        h0 = self._new(B, T, 320)
        ops.est_pack(x, mu, spks, cond, h0, B=B, T=T, dtype=dt, x_bstride=x_bstride, x_mod=(x_mod or B))
        xs = self._new(B, T, C, f32=True)
        cat = self._new(B, T, 2 * C)                     # [mid output | skip] for the up block
        # down block: resnet + 4 transformer blocks; the last block drops its activation copy into cat[:, :, C:]
        self._resnet(self.down["res"], h0, 320, B, T, tv, mask, xs)
        for j, w in enumerate(self.down["blocks"]):
            last = j == 3
            self._tblock(w, xs, B, T, mask, chunk, act_out=(cat[:, :, C:] if last else None), act_ld=2 * C, klen=klen)
        a = self._new(B, T, C)
        ops.gemm(cat[:, :, C:], self.down_w, T, C, dtype=dt, lda=2 * C, cin=C, ntaps=3, row_off=-2, row_lo=0, row_hi=T,
                 batch=B, a_bstride=T * 2 * C, bias=self.down_b, rowmask=mask, rm_bstride=T, out_act=a, ldo_a=C,
                 oa_bstride=T * C)
        lda = C
        for i, st in enumerate(self.mid):
            self._resnet(st["res"], a, lda, B, T, tv, mask, xs)
            lastst = i == len(self.mid) - 1
            for j, w in enumerate(st["blocks"]):
                last = j == 3
                if last and lastst:
                    self._tblock(w, xs, B, T, mask, chunk, act_out=cat, act_ld=2 * C, klen=klen)
                elif last:
                    self._tblock(w, xs, B, T, mask, chunk, act_out=a, act_ld=C, klen=klen)
                else:
                    self._tblock(w, xs, B, T, mask, chunk, klen=klen)
        self._resnet(self.up["res"], cat, 2 * C, B, T, tv, mask, xs)
        for j, w in enumerate(self.up["blocks"]):
            self._tblock(w, xs, B, T, mask, chunk, act_out=(a if j == 3 else None), act_ld=C, klen=klen)
        a2 = self._new(B, T, C)
        ops.conv1d(a, self.up_w, T=T, Cin=C, k=3, pad_left=2, dtype=dt, batch=B, bias=self.up_b, rowmask=mask, out_act=a2)
        c1 = self._new(B, T, C, f32=True)
        ops.conv1d(a2, self.fin_w, T=T, Cin=C, k=3, pad_left=2, dtype=dt, batch=B, bias=self.fin_b, out_f32=c1)
        ops.rownorm(c1, self.fin_g, self.fin_be, 1e-5, rows=T, C_=C, batch=B, act="mish", rowmask=mask, out_act=a, dtype=dt)
        if out is None:
            out = self._new(B, T, 80, f32=True)
        ops.conv1d(a, self.proj_w, T=T, Cin=C, k=1, dtype=dt, batch=B, bias=self.proj_b, rowmask=mask, out_f32=out)
        return out

    # ------------------------------------------------------------------ estimator on the row-tile fused kernels
    # measured time (us, bf16) of ONE launch of est_tail_kernel with few workgroups, by tile height (tools/tail_lab.py: 16 rows
    # x 8 waves, 32 and 64 rows x 8 waves with 32-column passes): every workgroup streams the block's 2 MB of weights from L2
    # through its CU's memory pipe, so a workgroup takes this long whatever else runs; with more of the chip streaming the
    # same weights it takes up to 45 % longer (256 workgroups: 34 / 42 / 65 us)
    _WG_US = {16: 27.1, 32: 29.9, 64: 44.1}
    # the same for the split build (two MFMAs per weight fragment; 64 rows: the K-halved tile of est_tail_tile, 8 waves):
    # profiles/r04_tail_lab64_x.txt - 1 000 rows: 29.1 / 35.8 / 60.3; 256 workgroups: - / 53.5 / 89.8
    _WG_US_X = {16: 29.1, 32: 35.8, 64: 60.3}
    # ... and with weight planes (an fp32-kind checkpoint: three MFMAs per fragment pair, twice the stream): profiles/r04_tail_lab64_xw.txt
    _WG_US_XW = {16: 46.9, 32: 52.6, 64: 81.2}

    @classmethod
    def _launch_us(cls, bm, tiles, split=False):
        """Model of one est_tail launch: full rounds of 256 workgroups, then the remainder.  split: False (bf16 build), True
        (split build) or "w" (split build with weight planes)."""
        wg = (cls._WG_US_XW if split == "w" else cls._WG_US_X if split else cls._WG_US)[bm]
        t = lambda n: wg * (1.0 + 0.45 * (n / 256.0) ** 2)
        full, rem = divmod(tiles, 256)
        return full * t(256) + (t(rem) if rem else 0.0)

    def _tile_rows(self, B, T):
        """Rows per workgroup of the fused kernels: the tile height with the shortest launch on 256 CUs."""
        tiles = lambda bm: B * ((T + bm - 1) // bm)
        if self.dtype == BF16:
            cap = getattr(self, "max_tile_rows", 64)
            bm = min((b for b in (64, 32, 16) if b <= cap), key=lambda b: (self._launch_us(b, tiles(b)), -b))
            bm = max(bm, getattr(self, "min_tile_rows", 16), 64 if self.polite else 16)
            return bm, bm
        if self.split:
            # two bf16 planes per LDS tile: the ResNet kernel's largest tile is 32 rows (the 512-channel ResNet of the up block:
            # 16, see _estimator_fused); the tail kernel has a 64-row form (attention tile in K halves, 256-wide FF chunks) whose
            # MFMA stages run at the MFMA's rate instead of the weight stream's: the tile with the shortest launch, and 64 rows
            # (half the workgroups of the 32-row tile at 0.85 of its time per row) beside the decode loop
            cap = getattr(self, "max_tile_rows", 64)
            bm = min((b for b in (64, 32, 16) if b <= cap), key=lambda b: (self._launch_us(b, tiles(b), "w" if self.wplanes else True), -b))
            if self.polite:
                bm = max(bm, min(cap, 64))
            br = 32 if (tiles(32) >= 128 or self.polite) else 16
            return bm, br
        return (32 if tiles(32) >= 128 else 16), 16            # fp32: tail, resnet (LDS: fp32 tiles are twice as large)

    def _estimator_fused(self, x, x_bstride, mu, spks, cond, t, B, T, mask, streaming, out, x_mod, klen=None):
        dt, C = self.dtype, self.C
        chunk = self.est_chunk if streaming else 0
        bm_t, bm_r = self._tile_rows(B, T)
        te = self._new(B, self.tdim)
        ops.sinusoidal_emb(t, te, dim=self.tdim, dtype=dt)
        t1 = self._new(B, 1024)
        ops.linear(te, self.t_w1, self.tdim, dtype=dt, bias=self.t_b1, act="silu", out_act=t1)
        t2 = self._new(B, 1024)
        ops.linear(t1, self.t_w2, 1024, dtype=dt, bias=self.t_b2, act2="mish", out_act=t2)
        ntv = self.mlp_w.shape[0]
        tv = self._new(B, ntv, f32=True)
        ops.linear(t2, self.mlp_w, 1024, dtype=dt, bias=self.mlp_b, out_f32=tv)
        h0 = self._new(B, T, 320)
        ops.est_pack(x, mu, spks, cond, h0, B=B, T=T, dtype=dt, x_bstride=x_bstride, x_mod=(x_mod or B))
        xs = self._new(B, T, C, f32=True)
        cat = self._new(B, T, 2 * C)
        a = self._new(B, T, C)
        ao = self._new(B, T, 512)
        bf = dt == BF16
        if bf:
            Tp = ops.round_up(T, 8)
            qk, vt = self._new(B, T, 1024), self._vt_buf(B, Tp)
            vt_bs = 512 * Tp
        elif self.split:
            # the producer splits the attention operands once: bf16 [hi Q | hi K | lo Q | lo K] rows and V^T planes
            Tp = ops.round_up(T, 8)
            qk = torch.empty(B, T, 2048, dtype=torch.bfloat16, device=self.dev)
            vt = self._vt_buf(B, Tp, planes=2)
            vt_bs = 2 * 512 * Tp
        else:
            qk, vt, Tp, vt_bs = self._new(B, T, 1536), None, 0, 0
        ldq = qk.shape[-1]

        def nxt(w):
            return ops.est_next(wqkv=w["wqkv_p"], n1g=w["n1g"], n1b=w["n1b"], q_out=qk, ldq=ldq, q_bs=T * ldq, vt_out=vt, ldvt=Tp,
                                vt_bs=vt_bs)

        def attention():
            if bf:
                ops.attn_flash_bf16(qk, qk[:, :, 512:], vt, ao, B=B, H=8, T=T, ldq=1024, ldk=1024, ldvt=Tp, ldo=512, q_bs=T * 1024,
                                    k_bs=T * 1024, vt_bs=512 * Tp, o_bs=T * 512, scale=0.125,
                                    keymask=(None if klen is not None else mask), chunk=chunk, fp8=self.attn_fp8, klen=klen)
            elif self.split:
                ops.attn_flash_xs(qk, vt, ao, B=B, H=8, T=T, ldqk=2048, ldvt=Tp, ldo=512, qk_bs=T * 2048, vt_bs=vt_bs, o_bs=T * 512,
                                  scale=0.125, keymask=(None if klen is not None else mask), chunk=chunk, klen=klen,
                                  form=(self.polite_flash_form if self.polite else 0))
            else:
                ops.attn_dense(qk, qk[:, :, 512:], qk[:, :, 1024:], ao, B=B, H=8, Tq=T, Tk=T, ldq=1536, ldk=1536, ldv=1536, ldo=512,
                               q_bs=T * 1536, k_bs=T * 1536, v_bs=T * 1536, o_bs=T * 512, scale=0.125, dtype=dt, keymask=mask,
                               chunk=chunk)

        def stage(st, a_in, lda, cin, act_out, act_ld):
            r, blocks = st["res"], st["blocks"]
            ops.est_resnet(a_in, lda, cin, xs, r, tv[:, r["idx"] * C:], ntv, B=B, T=T, dtype=dt,
                           bm=(16 if (self.split and cin > 320) else bm_r), rowmask=mask, nxt=nxt(blocks[0]))
            for j, w in enumerate(blocks):
                attention()
                last = j == len(blocks) - 1
                ops.est_tail(ao, xs, w, B=B, T=T, dtype=dt, bm=bm_t, rowmask=(mask if last else None),
                             act_out=(act_out if last else None), act_ld=act_ld, nxt=(None if last else nxt(blocks[j + 1])))

        # down block: its last transformer block drops the masked activation copy into cat[:, :, C:] (the skip)
        stage(self.down, h0, 320, 320, cat[:, :, C:], 2 * C)
        ops.gemm(cat[:, :, C:], self.down_w, T, C, dtype=dt, lda=2 * C, cin=C, ntaps=3, row_off=-2, row_lo=0, row_hi=T,
                 batch=B, a_bstride=T * 2 * C, bias=self.down_b, rowmask=mask, rm_bstride=T, out_act=a, ldo_a=C,
                 oa_bstride=T * C)
        for i, st in enumerate(self.mid):
            lastst = i == len(self.mid) - 1
            stage(st, a, C, C, cat if lastst else a, 2 * C if lastst else C)
        stage(self.up, cat, 2 * C, 2 * C, a, C)
        a2 = self._new(B, T, C)
        ops.conv1d(a, self.up_w, T=T, Cin=C, k=3, pad_left=2, dtype=dt, batch=B, bias=self.up_b, rowmask=mask, out_act=a2)
        c1 = self._new(B, T, C, f32=True)
        ops.conv1d(a2, self.fin_w, T=T, Cin=C, k=3, pad_left=2, dtype=dt, batch=B, bias=self.fin_b, out_f32=c1)
        ops.rownorm(c1, self.fin_g, self.fin_be, 1e-5, rows=T, C_=C, batch=B, act="mish", rowmask=mask, out_act=a, dtype=dt)
        if out is None:
            out = self._new(B, T, 80, f32=True)
        ops.conv1d(a, self.proj_w, T=T, Cin=C, k=1, dtype=dt, batch=B, bias=self.proj_b, rowmask=mask, out_f32=out)
        return out

    # ------------------------------------------------------------------ streaming with cached state (BASELINE config 5)
    class StreamState:
        """What a streaming utterance keeps between hops so that a hop solves only its NEW frames.  The estimator is
        chunk-causal in streaming mode (attention: query frame i sees keys < (i // 50 + 1) * 50, flow/decoder.py:441-445;
        every conv is causal), its inputs for finished frames never change (the encoder is chunk-causal with a 3-token
        look-ahead, the noise is fixed), and a hop adds exactly one 50-frame chunk.  So the trajectory of finished frames
        is final, and per Euler step s and per layer only two things of theirs are ever read again: the K / V rows of every
        transformer block and the last rows of every causal conv's input.  Both are kept, per Euler step, in buffers
        indexed by absolute frame; the kernels take `t_begin` / `q_begin` and compute only rows from there on.
        The reference recomputes every frame at every hop (cli/model.py:341-352): O(n^2) estimator work per utterance;
        with the cache it is O(n) plus attention over the cached keys.  Exact: tests/test_gpu_stream.py compares the two.
        Device memory per Euler step (bf16, B = 2): 56 blocks x (Q|K rows 4 KB + V^T 2 KB) + 16 conv inputs x ~0.6 KB per
        frame = 0.36 MB per frame -> 10.8 GB for a 60 s utterance (3000 frames, 10 steps); fp32 twice that."""

        def __init__(self, eng, Tcap):
            self.eng, self.Tcap, self.T = eng, ops.round_up(Tcap, 64), 0
            self.steps = [None] * eng.n_timesteps
            self.lat = torch.zeros(self.Tcap, 80, device=eng.dev)
            Tc, new = self.Tcap, eng._new
            self.xs, self.c1 = new(2, Tc, eng.C, f32=True), new(2, Tc, eng.C, f32=True)
            self.ao, self.afin, self.d = new(2, Tc, 512), new(2, Tc, eng.C), new(2, Tc, 80, f32=True)
            tt, dd = eng.t_schedule()
            self.t_all = torch.tensor([[v, v] for v in tt], dtype=torch.float32, device=eng.dev)
            self.dt = dd
            self.enc = None                                # encoder state (tokens), built on first use
            self.tok_done = 0                              # tokens encoded and solved
            self.z = eng.rand_noise[0, :, :self.Tcap].t().contiguous().to(eng.dev)      # [Tcap, 80] the fixed noise
            self.cond = torch.zeros(self.Tcap, 80, device=eng.dev)    # prompt latents on the prompt's frames, zero elsewhere
            self.spks2 = torch.zeros(2, 80, device=eng.dev)
            self.ids = torch.zeros(self.Tcap // 2 + eng.L + 8, dtype=torch.int64, device=eng.dev)
            self.graphs, self._win = {}, {}
            self.busy = False

        def release(self):
            """drops the recorded hop graphs and the buffers (an evicted state)"""
            for g in self.graphs.values():
                g.release()
            self.graphs, self._win, self.steps, self.enc = {}, {}, [None] * self.eng.n_timesteps, None

        def window(self, n):
            """static per-hop windows: ODE state of the n new frames, CFG pair of mu / cond rows (row 1 stays zero)"""
            if n not in self._win:
                d = self.eng.dev
                self._win[n] = dict(x=torch.zeros(1, n, 80, device=d), mu2=torch.zeros(2, n, 80, device=d),
                                    cond2=torch.zeros(2, n, 80, device=d))
            return self._win[n]

        def reset(self):
            """a new utterance on the same buffers and recorded hop graphs"""
            self.T = self.tok_done = 0

        def step_buffers(self, s):
            if self.steps[s] is None:
                e, Tc = self.eng, self.Tcap
                z = lambda *sh: torch.zeros(*sh, dtype=e.tdt, device=e.dev)
                nblk = 4 * (2 + len(e.mid))
                ldq = 1024 if e.dtype == BF16 else 1536
                self.steps[s] = dict(h0=z(2, Tc, 320), amid=[z(2, Tc, e.C) for _ in e.mid], cat=z(2, Tc, 2 * e.C), aup=z(2, Tc, e.C),
                                     a2=z(2, Tc, e.C), qk=[z(2, Tc, ldq) for _ in range(nblk)],
                                     vt=([z(2, 512, Tc) for _ in range(nblk)] if e.dtype == BF16 else [None] * nblk))
            return self.steps[s]

    # streaming state pool: capacities are bucketed coarsely (multiples of 512 frames, at most stream_cap_frames: beyond
    # the cap a hop falls back to the uncached whole-prefix solve, inference_time_major), idle states are kept for reuse
    # (their recorded hop graphs are the expensive part) under a byte budget, least recently used first out
    stream_cap_frames = 4096                        # 82 s of audio; 0.36 MB per frame in bf16 (all 10 Euler steps)
    stream_budget_bytes = 24 << 30

    def _stream_state_bytes(self, cap):
        per_frame = 0.36e6 * (1 if self.dtype == BF16 else 2)
        return int(cap * per_frame)

    def stream_open(self, max_frames: int) -> "FlowEngine.StreamState":
        """State for one streaming utterance of up to max_frames frames (capped at stream_cap_frames).  States are pooled
        per capacity bucket: buffers and the recorded per-hop graphs are reused by the next utterance of that bucket.  A state
        is busy from stream_open to stream_close (a second concurrent utterance of the same bucket gets its own state);
        idle states beyond stream_budget_bytes are dropped, least recently used first."""
        cap = min(ops.round_up(max(max_frames, 1), 512), ops.round_up(self.stream_cap_frames, 64))
        if not hasattr(self, "_stream_pool") or not isinstance(self._stream_pool, OrderedDict):
            self._stream_pool = OrderedDict()          # id -> state, in LRU order
        for key, st in self._stream_pool.items():
            if st.Tcap == cap and not st.busy:
                self._stream_pool.move_to_end(key)
                break
        else:
            self._stream_evict(self._stream_state_bytes(cap))
            st = FlowEngine.StreamState(self, cap)
            self._stream_pool[id(st)] = st
        st.busy = True
        st.reset()
        return st

    def stream_close(self, st):
        """The utterance is over: the state returns to the pool (or is dropped when the pool is over budget)."""
        st.busy = False
        self._stream_evict(0)

    def _stream_evict(self, incoming):
        pool = getattr(self, "_stream_pool", None)
        if not pool:
            return
        total = sum(self._stream_state_bytes(s_.Tcap) for s_ in pool.values()) + incoming
        for key in list(pool):
            if total <= self.stream_budget_bytes:
                break
            if not pool[key].busy:
                total -= self._stream_state_bytes(pool[key].Tcap)
                pool.pop(key).release()

    def _estimator_stream(self, st, s, x_new, mu_new, spks2, cond_new, tb, T):
        """One estimator call of Euler step s on frames tb .. T-1 of a streaming utterance (CFG pair, B = 2), reading the
        cached rows of earlier hops.  x_new [1, n, 80], mu_new / cond_new [2, n, 80] (row 1 zero), spks2 [2, 80].
        Returns d fp32 [2, Tcap, 80] (valid rows tb .. T-1)."""
        dt, C, Tc = self.dtype, self.C, st.Tcap
        S = st.step_buffers(s)
        B, n = 2, T - tb
        r0 = tb // 16 * 16                                   # tiles start on a 16-frame boundary: <= 15 finished frames are
        chunk = self.est_chunk                               # recomputed (same inputs, same values)
        bm_t, bm_r = self._tile_rows(B, T - r0)
        te = self._new(B, self.tdim)
        ops.sinusoidal_emb(st.t_all[s], te, dim=self.tdim, dtype=dt)
        t1 = self._new(B, 1024)
        ops.linear(te, self.t_w1, self.tdim, dtype=dt, bias=self.t_b1, act="silu", out_act=t1)
        t2 = self._new(B, 1024)
        ops.linear(t1, self.t_w2, 1024, dtype=dt, bias=self.t_b2, act2="mish", out_act=t2)
        ntv = self.mlp_w.shape[0]
        tv = self._new(B, ntv, f32=True)
        ops.linear(t2, self.mlp_w, 1024, dtype=dt, bias=self.mlp_b, out_f32=tv)
        h0w = self._new(B, n, 320)
        ops.est_pack(x_new, mu_new, spks2, cond_new, h0w, B=B, T=n, dtype=dt, x_bstride=n * 80, x_mod=1)
        S["h0"][:, tb:T].copy_(h0w)
        bf = dt == BF16
        blk = [0]

        def nxt(w, i):
            qk, vt = S["qk"][i], S["vt"][i]
            return ops.est_next(wqkv=w["wqkv_p"], n1g=w["n1g"], n1b=w["n1b"], q_out=qk, ldq=qk.shape[-1], q_bs=Tc * qk.shape[-1],
                                vt_out=vt, ldvt=(Tc if bf else 0), vt_bs=512 * Tc)

        def attention(i):
            qk, vt = S["qk"][i], S["vt"][i]
            if bf:
                ops.attn_flash_bf16(qk, qk[:, :, 512:], vt, st.ao, B=B, H=8, T=T, ldq=1024, ldk=1024, ldvt=Tc, ldo=512, q_bs=Tc * 1024,
                                    k_bs=Tc * 1024, vt_bs=512 * Tc, o_bs=Tc * 512, scale=0.125, chunk=chunk, q_begin=r0)
            elif self.split:
                ops.attn_flash_x(qk, qk[:, :, 512:], qk[:, :, 1024:], st.ao, B=B, H=8, T=T, ldq=1536, ldk=1536, ldv=1536, ldo=512,
                                 q_bs=Tc * 1536, k_bs=Tc * 1536, v_bs=Tc * 1536, o_bs=Tc * 512, scale=0.125, chunk=chunk, q_begin=r0)
            else:
                ops.attn_dense(qk, qk[:, :, 512:], qk[:, :, 1024:], st.ao, B=B, H=8, Tq=T, Tk=T, ldq=1536, ldk=1536, ldv=1536, ldo=512,
                               q_bs=Tc * 1536, k_bs=Tc * 1536, v_bs=Tc * 1536, o_bs=Tc * 512, scale=0.125, dtype=dt, chunk=chunk,
                               q_begin=r0)

        def stage(sw, a_in, lda, cin, act_out, act_ld):
            r, blocks = sw["res"], sw["blocks"]
            i0 = blk[0]
            ops.est_resnet(a_in, lda, cin, st.xs, r, tv[:, r["idx"] * C:], ntv, B=B, T=T, dtype=dt,
                           bm=(16 if (self.split and cin > 320) else bm_r), nxt=nxt(blocks[0], i0), t_begin=r0, Tcap=Tc)
            for j, w in enumerate(blocks):
                attention(i0 + j)
                last = j == len(blocks) - 1
                ops.est_tail(st.ao, st.xs, w, B=B, T=T, dtype=dt, bm=bm_t, act_out=(act_out if last else None), act_ld=act_ld,
                             nxt=(None if last else nxt(blocks[j + 1], i0 + j + 1)), t_begin=r0, Tcap=Tc)
            blk[0] += len(blocks)

        def conv3(src, ld, col0, cin, wgt, bias, out_act=None, out_f32=None, ldo=None):
            """causal conv k3 over frames r0 .. T-1 of a cached [2, Tcap, ld] buffer (rows before r0 are the halo)"""
            N = wgt.shape[0]
            # A is addressed from frame 0 of the buffer (the kernel's buffer descriptor cannot reach below its base):
            # output row m is frame r0 + m and reads frames r0 + m + tap - 2
            ops.gemm(src[:, :, col0:], wgt, T - r0, N, dtype=dt, lda=ld, cin=cin, ntaps=3, row_off=r0 - 2, row_lo=0, row_hi=T,
                     batch=B, a_bstride=Tc * ld, bias=bias, out_act=(out_act[:, r0:] if out_act is not None else None), ldo_a=N,
                     oa_bstride=Tc * N, out_f32=(out_f32[:, r0:] if out_f32 is not None else None), ldo_f=N, of_bstride=Tc * N)

        stage(self.down, S["h0"], 320, 320, S["cat"][:, :, C:], 2 * C)
        conv3(S["cat"], 2 * C, C, C, self.down_w, self.down_b, out_act=S["amid"][0])
        for i, sw in enumerate(self.mid):
            lastst = i == len(self.mid) - 1
            stage(sw, S["amid"][i], C, C, S["cat"] if lastst else S["amid"][i + 1], 2 * C if lastst else C)
        stage(self.up, S["cat"], 2 * C, 2 * C, S["aup"], C)
        conv3(S["aup"], C, 0, C, self.up_w, self.up_b, out_act=S["a2"])
        conv3(S["a2"], C, 0, C, self.fin_w, self.fin_b, out_f32=st.c1)
        ops.rownorm(st.c1[:, r0:], self.fin_g, self.fin_be, 1e-5, rows=T - r0, C_=C, batch=B, x_bstride=Tc * C, act="mish",
                    out_act=st.afin[:, r0:], o_bstride=Tc * C, dtype=dt)
        ops.gemm(st.afin[:, r0:], self.proj_w, T - r0, 80, dtype=dt, lda=C, cin=C, batch=B, a_bstride=Tc * C, bias=self.proj_b,
                 out_f32=st.d[:, r0:], ldo_f=80, of_bstride=Tc * 80)
        return st.d

    def _cfm_stream_rows(self, st, tb: int, T: int):
        """Euler-solves frames tb .. T-1 (whole 50-frame chunks) from the state's mu / speaker rows; the latents land in
        st.lat.  Pure device work on state buffers (recorded into the per-hop hipGraph)."""
        assert tb < T <= st.Tcap and tb % self.est_chunk == 0 and T % self.est_chunk == 0, (tb, T, st.Tcap)
        n = T - tb
        W = st.window(n)
        W["x"].copy_(st.z[tb:T].reshape(1, n, 80))
        W["mu2"][0].copy_(st.enc["mu"][tb:T])
        W["cond2"][0].copy_(st.cond[tb:T])
        for s in range(self.n_timesteps):
            d = self._estimator_stream(st, s, W["x"], W["mu2"], st.spks2, W["cond2"], tb, T)
            ops.cfg_euler(W["x"], d[0, tb:T], d[1, tb:T], self.cfg, st.dt[s], n * 80)
        st.lat[tb:T].copy_(W["x"][0])

    @torch.no_grad()
    def stream_hop(self, st, ids: torch.Tensor, embedding: torch.Tensor, prompt_feat: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One streaming hop: ids [Lt] = (prompt tokens ++) all tokens so far + 3 look-ahead tokens.  Encodes and solves only
        what the state has not seen; each (done, Lt) pair is one recorded hipGraph over the state's buffers (a hop is
        ~1 500 short launches), reused by every later utterance that runs on this state.  prompt_feat [Tp, 80] (first hop of
        a zero-shot utterance, flow.py:472-498): the latents of the prompt's frames, the `cond` rows of those frames.
        Returns latents fp32 [2(Lt-3), 80] of ALL frames (prompt frames first)."""
        Lt = ids.numel()
        T, tb = Lt - self.L, st.tok_done
        if tb == 0:                                          # speaker projection and prompt condition, once per utterance
            st.cond.zero_()
            if prompt_feat is not None and prompt_feat.shape[0]:
                st.cond[:prompt_feat.shape[0]].copy_(prompt_feat.to(self.dev, torch.float32))
            st.spks2.zero_()
            en = self._new(1, self.spk_dim)
            ops.rownorm(embedding.to(self.dev, torch.float32).contiguous(), self.spk_gamma, None, 1e-30, rows=1, C_=self.spk_dim,
                        rms=True, out_act=en, dtype=self.dtype)
            ops.linear(en, self.spk_w, self.spk_dim, dtype=self.dtype, bias=self.spk_b, out_f32=st.spks2[0:1])
        self._enc_stream_state(st)
        st.ids[:Lt].copy_(ids.reshape(-1))
        key = (tb, Lt)
        if key not in st.graphs:
            def hop(tb=tb, Lt=Lt, T=T):
                self._encode_stream_rows(st, st.ids[:Lt], tb, Lt)
                self._cfm_stream_rows(st, 2 * tb, 2 * T)
            st.graphs[key] = Graphed(hop, self.use_graphs)
        st.graphs[key]()
        st.tok_done = T
        st.T = 2 * T
        return st.lat[:2 * T]

    @torch.no_grad()
    def estimator_channels_first(self, x, mask, mu, t, spks, cond, streaming=False):
        """The reference's estimator seam (flow_matching.py:128-131; ONNX names x, mask, mu, t, spks, cond):
        x/mu/cond [B,80,T], mask [B,1,T], t [B], spks [B,80] fp32 -> [B,80,T] fp32 (a fresh tensor)."""
        B, _, T = x.shape
        tm = lambda a: self._to_time_major(a.to(self.dev, torch.float32).contiguous(), B, T)
        m = mask.to(self.dev, torch.float32).reshape(B, T).contiguous()
        d = self.estimator(tm(x), T * 80, tm(mu), spks.to(self.dev, torch.float32).contiguous(), tm(cond),
                           t.to(self.dev, torch.float32).contiguous(), B, T, m, streaming)
        out = torch.empty(B, 80, T, dtype=torch.float32, device=self.dev)
        ops.copy2d(d, F32, T * 80, 80, 1, out, F32, 80 * T, 1, T, rows=T, cols=80, batch=B)
        return out

    def _to_time_major(self, a, B, T):
        o = torch.empty(B, T, 80, dtype=torch.float32, device=self.dev)
        ops.copy2d(a, F32, 80 * T, 1, T, o, F32, T * 80, 80, 1, rows=T, cols=80, batch=B)
        return o

    # ------------------------------------------------------------------ CFM solve
    def t_schedule(self):
        """flow_matching.py:345-348 + :90,121-124 replicated in fp32 on the host: (t per step, dt per step)."""
        n = self.n_timesteps
        ts = torch.linspace(0, 1, n + 1, dtype=torch.float32)
        ts = 1 - torch.cos(ts * 0.5 * torch.pi)
        t, dt = ts[0:1].clone(), ts[1] - ts[0]
        tt, dd = [], []
        for step in range(1, n + 1):
            tt.append(float(t))
            dd.append(float(dt))
            t = t + dt
            if step < n:
                dt = ts[step + 1] - t
        return tt, dd

    class _Plan:
        pass

    def _cfm_plan(self, n, T, streaming, masked):
        """Plan for n utterances padded to T frames: static buffers + the graph of the whole Euler solve."""
        key = ("cfm", n, T, bool(streaming), bool(masked), bool(self.polite))
        if key in self._plans:
            self._plans.move_to_end(key)
            return self._plans[key]
        P = FlowEngine._Plan()
        # A plan owns its static buffers and (once recorded) a hipGraph of ~5 000 nodes with a private activation pool:
        # about 2n*T*16 KB of device memory.  Plans are kept in an LRU under `plan_budget_bytes`; an evicted plan's
        # graph and pool are released, a later call of that shape records it again (one eager pass + one capture).
        P.nbytes = 2 * n * T * (16 << 10)
        while self._plans and self.plan_bytes + P.nbytes > self.plan_budget_bytes:
            _, old = self._plans.popitem(last=False)
            self._release_plan(old)
        self.plan_bytes += P.nbytes
        P.x = self._new(n, T, 80, f32=True)             # ODE state, shared by the two halves of the CFG batch
        P.mu = torch.zeros(2 * n, T, 80, device=self.dev)   # rows n.. stay zero: the unconditional branch
        P.spks = torch.zeros(2 * n, 80, device=self.dev)
        P.cond = torch.zeros(2 * n, T, 80, device=self.dev)
        P.mask = torch.ones(2 * n, T, device=self.dev) if masked else None
        # the masks of a padded group are prefixes: the flash kernel takes the lengths (no masked tiles inside an utterance,
        # no key tiles beyond it) instead of the mask
        P.klen = torch.full((2 * n,), T, dtype=torch.int32, device=self.dev) if masked else None
        P.d = self._new(2 * n, T, 80, f32=True)
        tt, dd = self.t_schedule()
        P.t_all = torch.tensor([[v] * (2 * n) for v in tt], dtype=torch.float32, device=self.dev)
        P.z = self.rand_noise[0, :, :T].t().contiguous().to(self.dev)      # [T,80], the same noise for every utterance

        def run():
            P.x.copy_(P.z.unsqueeze(0).expand(n, T, 80))
            for s in range(self.n_timesteps):
                self.estimator(P.x, T * 80, P.mu, P.spks, P.cond, P.t_all[s], 2 * n, T, P.mask, streaming, out=P.d, x_mod=n, klen=P.klen)
                ops.cfg_euler(P.x, P.d[:n], P.d[n:], self.cfg, dd[s], n * T * 80)

        P.run = Graphed(run, self.use_graphs)
        self._plans[key] = P
        return P

    def _release_plan(self, P):
        """A plan is a reference cycle (P.run's closure holds P), so dropping the last name would leave its hipGraph and
        its private pool to the cyclic collector: release them here, now, on this thread, and break the cycle."""
        self.plan_bytes -= P.nbytes
        P.run.release()
        P.__dict__.clear()

    def close(self):
        """Deterministic teardown: every recorded graph (Euler-solve plans, streaming hop graphs) is destroyed on the
        calling thread.  The engine can still be used afterwards (plans are recorded again on demand)."""
        while self._plans:
            _, old = self._plans.popitem(last=False)
            self._release_plan(old)
        pool = getattr(self, "_stream_pool", None) or {}
        for key in list(pool):
            pool.pop(key).release()

    def cfm_batch(self, mus, spks, conds, streaming=False, pad_to=1):
        """n utterances in one solve: mus/conds lists of fp32 [T_i,80], spks list of [80].  Shorter utterances are
        zero padded and masked (row mask on every activation, key mask in attention), exactly like a padded batch
        of the reference's estimator (decoder.py:433-496 with mask).  Returns a list of fp32 [T_i,80] views of the
        plan's state (copy before the next call)."""
        n = len(mus)
        Ts = [m.shape[0] for m in mus]
        T = ops.round_up(max(Ts), pad_to)
        masked = any(t != T for t in Ts)
        if self.shape_log is not None:                     # measurement hook (bench.py): group shapes of a run
            self.shape_log.append((n, T, sum(Ts), sum(t * t for t in Ts), bool(self.polite)))
        P = self._cfm_plan(n, T, streaming, masked)
        if masked:
            P.mu[:n].zero_()
            P.cond[:n].zero_()
            P.mask.zero_()
        for i in range(n):
            P.mu[i, :Ts[i]].copy_(mus[i])
            P.cond[i, :Ts[i]].copy_(conds[i])
            P.spks[i].copy_(spks[i].reshape(-1))
            if masked:
                P.mask[i, :Ts[i]] = 1.0
                P.mask[n + i, :Ts[i]] = 1.0
        if masked:
            P.klen.copy_(torch.tensor(list(Ts) * 2, dtype=torch.int32), non_blocking=False)
        P.run()
        return [P.x[i, :Ts[i]] for i in range(n)]

    def cfm(self, mu, spks, cond, streaming=False) -> torch.Tensor:
        """mu, cond fp32 [T,80] time-major; spks fp32 [80] -> x fp32 [T,80] (owned by the plan: copy if kept)."""
        # streaming hops ask for a new length every time: bucket T to two chunks (the row / key masks make the padded
        # solve equal to the unpadded one), so a 60 s utterance records 30 plans instead of 60
        return self.cfm_batch([mu], [spks], [cond], streaming, pad_to=(2 * self.est_chunk if streaming else 1))[0]

    # ------------------------------------------------------------------ flow.inference
    @torch.no_grad()
    def inference_time_major(self, token, prompt_token, prompt_feat, embedding, streaming=False, finalize=True,
                             reference_mels=None, stream_state=None):
        """token [1,Lt], prompt_token [1,Lp] ints; prompt_feat [1,Tp,80]; embedding [1,192] (device tensors).
        Returns fp32 [T2, 80] time-major latents of the NEW tokens (prompt part dropped).  stream_state (a StreamState from
        stream_open, streaming non-final calls): only the frames that state has not solved yet go through the ODE."""
        Lt, Lp, Tp = token.numel(), prompt_token.numel(), prompt_feat.shape[1]
        La = Lp + Lt                                         # the flow runs over prompt ++ tokens (flow.py:472-476)
        if (stream_state is not None and streaming and not finalize and Tp == 2 * Lp
                and reference_mels is None and (2 * (La - self.L)) % self.est_chunk == 0 and 2 * (La - self.L) <= stream_state.Tcap
                and stream_state.tok_done < La - self.L):
            ids = torch.cat([prompt_token.reshape(-1), token.reshape(-1)]).to(self.dev, torch.int64)
            return self.stream_hop(stream_state, ids, embedding, prompt_feat[0] if Tp else None)[Tp:]
        mu, spks, cond, mel_len1 = self.conditions(token, prompt_token, prompt_feat, embedding, streaming, finalize, reference_mels)
        x = self.cfm(mu, spks, cond, streaming)
        return x[mel_len1:]

    @torch.no_grad()
    def conditions(self, token, prompt_token, prompt_feat, embedding, streaming=False, finalize=True, reference_mels=None):
        """Everything of flow.inference ahead of the ODE solve (flow.py:455-498): speaker projection, token
        embedding + conformer encoder -> mu, prompt condition.  Returns (mu [T,80], spks [1,80], cond [T,80], Tp).
        reference_mels ([1,N,80,T] or [1,80,T]) selects the learnable speaker encoder (flow.py:456-462)."""
        dt = self.dtype
        if reference_mels is not None and self.spk_enc is not None:
            emb = self.spk_enc.reference_embedding(reference_mels)
        else:
            emb = embedding.to(self.dev, torch.float32).contiguous()
        en = self._new(1, self.spk_dim)
        # F.normalize(embedding, dim=1) == rmsnorm with gamma 1/sqrt(d) and eps -> 0
        ops.rownorm(emb, self.spk_gamma, None, 1e-30, rows=1, C_=self.spk_dim, rms=True, out_act=en, dtype=dt)
        spks = self._new(1, 80, f32=True)
        ops.linear(en, self.spk_w, self.spk_dim, dtype=dt, bias=self.spk_b, out_f32=spks)
        ids = torch.cat([prompt_token.reshape(-1), token.reshape(-1)]).to(self.dev, torch.int64)
        mu = self.encode(ids, finalize, streaming)
        T = mu.shape[0]
        mel_len1 = prompt_feat.shape[1]
        cond = torch.zeros(T, 80, device=self.dev)
        if mel_len1:
            cond[:mel_len1].copy_(prompt_feat[0].to(self.dev, torch.float32))
        return mu, spks, cond, mel_len1

    def conditions_batch(self, tokens, prompt_tokens, prompt_feats, embeddings):
        """conditions() for several whole utterances (finalize, not streaming), the conformer encoder batched over them
        (encode_batch).  Returns the list of (mu, spks, cond, Tp) conditions() would."""
        dt, n = self.dtype, len(tokens)
        emb = torch.cat([e.to(self.dev, torch.float32).reshape(1, -1) for e in embeddings], 0).contiguous()
        en = self._new(n, self.spk_dim)
        ops.rownorm(emb, self.spk_gamma, None, 1e-30, rows=n, C_=self.spk_dim, rms=True, out_act=en, dtype=dt)
        spks = self._new(n, 80, f32=True)
        ops.linear(en, self.spk_w, self.spk_dim, dtype=dt, bias=self.spk_b, out_f32=spks)
        ids = [torch.cat([p.reshape(-1), t.reshape(-1)]).to(self.dev, torch.int64) for p, t in zip(prompt_tokens, tokens)]
        mus = self.encode_batch(ids)
        out = []
        for b in range(n):
            T, L1 = mus[b].shape[0], prompt_feats[b].shape[1]
            cond = torch.zeros(T, 80, device=self.dev)
            if L1:
                cond[:L1].copy_(prompt_feats[b][0].to(self.dev, torch.float32))
            out.append((mus[b], spks[b:b + 1], cond, L1))
        return out

    @torch.no_grad()
    def inference(self, token, prompt_token, prompt_feat, embedding, streaming=False, finalize=True, reference_mels=None):
        """Reference layout: returns feat [1, 80, T2] fp32 (flow.py:509-511)."""
        x = self.inference_time_major(token, prompt_token, prompt_feat, embedding, streaming, finalize, reference_mels)
        T2 = x.shape[0]
        out = torch.empty(1, 80, T2, dtype=torch.float32, device=self.dev)
        ops.copy2d(x, F32, 0, 80, 1, out, F32, 0, 1, T2, rows=T2, cols=80)
        return out
