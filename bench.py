"""bench.py — headline benchmark of the TTS hot path (BASELINE.json metric: 24 kHz audio-seconds per wall-second
per node, + RTF), CosyVoice2-0.5B-shaped models, random-init weights, synthetic inputs.

One "step" = one full pass of the path over the rank's batch: AR LM decode (text ids -> FSQ tokens), flow-matching
decoder (tokens -> latents, 10 Euler steps, CFG), DAC-VAE decoder (latents -> 24 kHz waveform) and, for N > 1, the
RCCL all-gather of the generated audio.  Inputs are resident on the device before the timed region.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "minimax-speech_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFS = 2500.0    # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 MFMA
MFMA_F32_PEAK_TFS = 157.3      # f32-input MFMA = the fp32 vector rate


CKPT_KIND = "bf16"


def build_weights(seed=0, kind=None):
    from mmx import shapes, synth
    kind = kind or CKPT_KIND
    return (synth.synth_state_dict(shapes.llm_manifest(), seed, kind=kind), synth.synth_state_dict(shapes.flow_manifest(), seed, kind=kind),
            synth.synth_state_dict(shapes.dac_decoder_manifest(80), seed, kind=kind))


def measure_lm_kernel(eng, iters=240):
    """Roofline of the LM decode step's largest projection: gate/up + SwiGLU (17.4 MB of the 727.6 MB of weights a step
    streams) on the decode-step projection kernel (csrc/decode.hip: skinny3_kernel) at the workload's batch size.
    Timed live with HIP events on the stream the kernel is launched on, a hipGraph of `iters` launches rotating over the 24
    layers' weights so that the 256 MiB Infinity Cache cannot serve the stream.  Algorithmic bytes per launch (SURVEY.md 8d:
    bf16 weights are the traffic of a decode step) = the packed bf16 weight matrix 2 x 4864 x 896 x 2 B + the activation
    planes in and out."""
    from mmx import ops
    llm = eng.llm
    B, H, I = llm.B, llm.H, llm.I
    S = llm._planes()
    ns = S["xs_b"].shape[0]
    nbytes = 2 * I * H * 2 + ns * (B * H + B * I) * 2
    run = lambda l: ops.skinny2(S["xs_b"], llm.layers[l % llm.n_layers]["wgu"], B=B, K=H, N=I, dtype=llm.ddt, ssq_in=S["ssq_b"],
                                eps=llm.eps, epi=1, xs_out=S["xs_act"], tiles_per_wg=llm.v2_cfg["gu"][0])
    us = _event_time_graph(run, iters)
    achieved = nbytes / (us * 1e-6) / 1e9
    return {"bound": "hbm", "kernel": f"skinny3_kernel (LM gate/up + SwiGLU, K=896, N=2x4864, batch {B}, {ns} {'fp16' if getattr(llm, 'h2', False) else 'bf16'} activation plane(s))",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": _pmc("lm_gate_up_hbm_bytes_per_launch", ns > 1), "bytes_per_launch": nbytes, "us_per_launch": round(us, 3)}


def measure_lm_step(eng, ctx=300, iters=96):
    """HBM roofline of ONE WHOLE decode step of the LM at the workload's batch size - the critical path of the step (the
    decode loop is 85 % of it): the captured decode step (every projection, the decode attention, the head, the sampler)
    replayed `iters` times from context length `ctx`, timed with HIP events on the stream the graph replays on.
    Algorithmic bytes per step: the bf16 weights once per step, 715.8 MB backbone + 11.8 MB llm_decoder =
    727.6 MB whatever the batch, + the KV cache rows read, batch * ctx * 2 (K, V) * 2 kv heads * 64 * 24 layers * element
    size (the split build keeps the cache in fp32)."""
    llm = eng.llm
    B = llm.B
    z = torch.zeros(1, 0, dtype=torch.long, device=llm.dev)
    g = torch.Generator().manual_seed(11)
    xs = [llm.build_lm_input(torch.randint(0, 151936, (1, 48), generator=g).to(llm.dev), z, z) for _ in range(B)]
    n = ctx - 50 + iters + 16
    llm.start(xs, [n] * B, [n] * B, seed=0)
    for _ in range(ctx - 50):                               # the first call of a new graph key runs eagerly, the second records
        llm.step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(torch.cuda.current_stream())
    for _ in range(iters):
        llm.step()
    e1.record(torch.cuda.current_stream())
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    kv_esz = 2 if llm.tdt == torch.bfloat16 else 4
    wbytes = 727.6e6 if llm.layers[0]["wqkv"].dtype == torch.bfloat16 else 2 * 727.6e6
    nbytes = wbytes + B * (ctx + iters / 2) * 2 * llm.Hkv * llm.D * llm.n_layers * kv_esz
    achieved = nbytes / (us * 1e-6) / 1e9
    return {"bound": "hbm", "kernel": f"one captured LM decode step at batch {B}, context {ctx}..{ctx + iters} ({llm.n_layers} layers x 5 launches + head + sampler)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": _pmc("lm_step_hbm_bytes", llm.tdt != torch.bfloat16), "bytes_per_launch": int(nbytes), "us_per_launch": round(us, 2),
            "launches_per_step": 5 * llm.n_layers + 3}


def _pmc(key, split=False):
    """HBM bytes per launch from the committed PMC passes (profiles/r04_pmc.json, else r03_pmc.json: FETCH_SIZE x 2 +
    WRITE_SIZE, collected and corrected as MI355X_MICROARCH.md prescribes, one counter per rocprofv3 pass; keys of the split
    build end in _x); None when no file holds the key."""
    for name in ("r04_pmc.json", "r03_pmc.json"):
        pj = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pj):
            v = json.load(open(pj)).get(key + ("_x" if split else ""))
            if v is not None:
                return v
    return None


def _event_time_graph(fn, iters):
    """Average duration (us) of `fn`'s launches: recorded `iters` times into one hipGraph (the launch mechanism of the
    pipeline), replayed once untimed and once between two HIP events on the launch stream."""
    fn(0)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    try:
        with torch.cuda.graph(g):
            for i in range(iters):
                fn(i)
    finally:
        gc.enable()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    g.replay()
    e1.record(torch.cuda.current_stream())
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def _time_est_tail(fl, n, T, polite=False):
    """Average duration (us) of one est_tail launch at group shape (n utterances x T frames, CFG pair -> M = 2nT rows):
    a hipGraph of 112 launches rotating over the 56 mid blocks' weights (2 MB each, as in the pipeline), timed with HIP
    events on the launch stream.  Returns (us, rows per workgroup)."""
    from mmx import ops
    dt = fl.dtype
    B, C = 2 * n, fl.C
    blocks = [w for st in fl.mid for w in st["blocks"]]
    Tp = ops.round_up(T, 8)
    ao = torch.randn(B, T, 512, device=fl.dev).to(fl.tdt)
    x = torch.randn(B, T, C, device=fl.dev)
    if fl.split:                                        # pre-split attention operands: [hi Q | hi K | lo Q | lo K] rows, two V^T planes
        qk, vt = torch.empty(B, T, 2048, dtype=torch.bfloat16, device=fl.dev), torch.zeros(B, 2, 512, Tp, dtype=torch.bfloat16, device=fl.dev)
    else:
        qk, vt = fl._new(B, T, 1024), torch.zeros(B, 512, Tp, dtype=fl.tdt, device=fl.dev)
    ldq, vt_bs = qk.shape[-1], vt[0].numel()
    was, fl.polite = fl.polite, polite                  # the tiling the step used for this group (FlowEngine.polite)
    bm, _ = fl._tile_rows(B, T)
    fl.polite = was

    def one(i=0):
        w, wn = blocks[i % len(blocks)], blocks[(i + 1) % len(blocks)]
        nxt = ops.est_next(wqkv=wn["wqkv_p"], n1g=wn["n1g"], n1b=wn["n1b"], q_out=qk, ldq=ldq, q_bs=T * ldq, vt_out=vt,
                           ldvt=Tp, vt_bs=vt_bs)
        ops.est_tail(ao, x, w, B=B, T=T, dtype=dt, bm=bm, nxt=nxt)

    return _event_time_graph(one, 2 * len(blocks)), bm


def measure_attn_kernel(eng, shapes, steps=1):
    """The estimator's flash attention (attn_flash_kernel, as large a share of the step's GPU time as est_tail_kernel) over
    the step's own flow-group shapes, measured like est_tail: a hipGraph of 56 launches per shape, HIP events on the launch
    stream, launch-weighted.  Algorithmic FLOPs per launch = 4 * 64 * 8 heads * 2 (CFG pair) * sum_i T_i^2 over the group's
    utterances (Q K^T and P V of the valid frames; padded rows and masked keys are not counted)."""
    from mmx import ops
    fl = eng.flow
    count = {}
    tot_us = tot_fl = tot_n = 0.0
    for n, T, _, sq, pol in shapes:
        count.setdefault((n, T, bool(pol)), [0, 0])[0] += 1
        tot_fl += 4.0 * 64 * 8 * 2 * sq
    for (n, T, pol), (cnt, _) in sorted(count.items()):
        B, Tp = 2 * n, ops.round_up(T, 8)
        form = fl.polite_flash_form if (pol and fl.split) else 0     # the workgroup shape the step used for this group (FlowEngine.polite)
        if fl.split:
            qk = torch.randn(B, T, 2048, device=fl.dev).to(torch.bfloat16)
            vt = torch.randn(B, 2, 512, Tp, device=fl.dev).to(torch.bfloat16)
        else:
            qk = torch.randn(B, T, 1024, device=fl.dev).to(fl.tdt)
            vt = torch.randn(B, 512, Tp, device=fl.dev).to(fl.tdt)
        ao = torch.empty(B, T, 512, device=fl.dev, dtype=fl.tdt)
        # the group as the step ran it: utterance lengths spread evenly from the group's mean down and up to T (the log
        # keeps n, T, sum T_i, sum T_i^2, not every length), passed as klen like FlowEngine does for a padded group
        valid = sum(v for n_, T_, v, _, p_ in shapes if (n_, T_, bool(p_)) == (n, T, pol)) / cnt
        lo = max(1, int(2 * valid / n - T))
        lens = [int(round(lo + (T - lo) * i / max(1, n - 1))) for i in range(n)]
        lens[-1] = T
        klen = torch.tensor(lens * 2, dtype=torch.int32, device=fl.dev)

        def one(i=0):
            if fl.split:
                ops.attn_flash_xs(qk, vt, ao, B=B, H=8, T=T, ldqk=2048, ldvt=Tp, ldo=512, qk_bs=T * 2048, vt_bs=2 * 512 * Tp, o_bs=T * 512,
                                  scale=0.125, klen=(klen if n > 1 else None), form=form)
            else:
                ops.attn_flash_bf16(qk, qk[:, :, 512:], vt, ao, B=B, H=8, T=T, ldq=1024, ldk=1024, ldvt=Tp, ldo=512, q_bs=T * 1024,
                                    k_bs=T * 1024, vt_bs=512 * Tp, o_bs=T * 512, scale=0.125, klen=(klen if n > 1 else None))

        tot_us += cnt * _event_time_graph(one, 56)
        tot_n += cnt
    us, flops = tot_us / tot_n, tot_fl / tot_n
    tfs = flops / (us * 1e-6) / 1e12
    traffic = _pmc("attn_flash_bench_hbm_bytes_per_launch", fl.split)
    kname = ("attn_flash_x_kernel (estimator attention, 8 heads x 64, split build: hi*hi + lo*hi + hi*lo = 3 bf16 MFMAs per product; `achieved` counts "
             "the algorithmic FLOPs once)") if fl.split else "attn_flash_kernel (estimator attention, 8 heads x 64, bf16)"
    return {"bound": "mfma", "kernel": f"{kname} over the {int(tot_n) // max(1, steps)} flow groups of a step",
            "achieved": round(tfs, 1), "peak": MFMA_BF16_PEAK_TFS, "unit": "TFLOP/s", "frac": round(tfs / MFMA_BF16_PEAK_TFS, 4),
            "traffic": traffic, "flops_per_launch": round(flops), "us_per_launch": round(us, 3)}


def measure_flow_kernel(eng, shapes, steps=1):
    """Roofline of the kernel family with the largest share of the step's GPU time (rocprofv3 kernel stats of this
    command under profiles/): est_tail_kernel, the row-tile fused kernel of the estimator's transformer blocks
    (attention-output projection + residual -> LayerNorm -> FF1 + GELU -> FF2 + residual -> LayerNorm -> Q/K/V
    projection of the next block).  It is launched 560 times per flow group, at M = 2 n T rows for a group of n
    utterances padded to T frames, so `achieved` is taken over the step's OWN launches: every distinct (n, T) of the
    step's flow groups is timed live (HIP events around a hipGraph of 112 launches on the launch stream) and weighted by
    how often the step launches it.  Algorithmic FLOPs per launch = 2 * (valid rows) * (512*256 + 256*1024 + 1024*256 +
    256*1536) (the linear part of a transformer block; padded rows are not counted); `us_per_launch`
    is the launch-weighted mean, comparable with the kernel's AverageNs in the committed rocprofv3 summary (there the
    kernels run beside the decode loop).  `isolated_large` is the same kernel on a chip-filling launch (M = 14 336)."""
    fl = eng.flow
    assert fl.dtype in (1, 2)
    per_row = 2.0 * (512 * fl.C + fl.C * 1024 + 1024 * fl.C + fl.C * 1536)
    count = {}
    for n, T, valid, _, pol in shapes:
        c = count.setdefault((n, T, pol), [0, 0])
        c[0] += 1
        c[1] += valid
    tot_us = tot_fast = tot_fl = tot_n = 0.0
    tiles = set()
    for (n, T, pol), (cnt, valid) in sorted(count.items()):
        us, bm = _time_est_tail(fl, n, T, pol)
        tiles.add(bm)
        tot_us += cnt * us
        tot_fast += cnt * (_time_est_tail(fl, n, T, False)[0] if pol else us)
        tot_fl += per_row * 2 * valid
        tot_n += cnt
    us, flops = tot_us / tot_n, tot_fl / tot_n
    tfs = flops / (us * 1e-6) / 1e12
    tfs_fast = flops / (tot_fast / tot_n * 1e-6) / 1e12
    us_l, bm_l = _time_est_tail(fl, 8, 896)
    tfs_l = per_row * 14336 / (us_l * 1e-6) / 1e12
    traffic, traffic_l = _pmc("est_tail_bench_hbm_bytes_per_launch", fl.split), _pmc("est_tail_8x896_hbm_bytes_per_launch", fl.split)
    bname = "split build: 2 bf16 MFMAs per weight fragment, `achieved` counts the algorithmic FLOPs once" if fl.split else "bf16"
    return {"bound": "mfma", "kernel": f"est_tail_kernel<{bname}, {'|'.join(str(t) for t in sorted(tiles))} rows per workgroup> (fused transformer-block tail) over the "
            f"{int(tot_n) // max(1, steps)} flow groups of a step ({len(count)} shapes, M = 2nT from {min(2 * n * T for n, T, _ in count)} to {max(2 * n * T for n, T, _ in count)} rows; "
            f"{sum(c[0] for k, c in count.items() if k[2]) // max(1, steps)} of them beside the decode loop, on the polite tiling)",
            "achieved": round(tfs, 1), "peak": MFMA_BF16_PEAK_TFS, "unit": "TFLOP/s", "frac": round(tfs / MFMA_BF16_PEAK_TFS, 4),
            "traffic": traffic, "flops_per_launch": round(flops), "us_per_launch": round(us, 3),
            "fastest_tiling": {"note": "the same launches with the tile height that minimises the launch itself (what the groups issued after the decode "
                               "loop use); beside the decode loop the step trades this for fewer workgroups", "achieved": round(tfs_fast, 1),
                               "frac": round(tfs_fast / MFMA_BF16_PEAK_TFS, 4), "us_per_launch": round(tot_fast / tot_n, 3)},
            "isolated_large": {"kernel": f"est_tail_kernel<{bm_l} rows>, M = 14336 (8 utterances x 896 frames x CFG pair)", "achieved": round(tfs_l, 1),
                               "frac": round(tfs_l / MFMA_BF16_PEAK_TFS, 4), "us_per_launch": round(us_l, 3), "traffic": traffic_l}}


def cpu_baseline():
    """The CPU oracle (port of the reference algorithm, oracle/*.py) timed on this host's cores on a bounded sample of
    the config-3 workload: per stage 1 warm-up + 3 repetitions, min and median — 16 LM decode steps
    after the 50-row prefill, the flow on 250 tokens (500 frames, 10 Euler steps, CFG), the DAC decoder on 500 frames.
    `value` = audio-seconds per second of a 10 s utterance built from the per-stage MINIMA (250 x token + flow + DAC)."""
    from oracle import dac as ODAC, flow as OFLOW, llm as OLLM
    # the GPU box gives a 1-GPU job a share of ~16 host cores; os.cpu_count() reports the whole machine
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    print(f"[cpu_baseline] timing the CPU oracle on {cores} threads ...", file=sys.stderr, flush=True)
    llm_sd, flow_sd, dac_sd = build_weights(0)
    g = torch.Generator().manual_seed(2)

    def reps(fn, n=3):
        fn()                                                   # warm-up
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[0], ts[len(ts) // 2]

    with torch.no_grad():
        cfg = OLLM.QwenCfg()
        x = OLLM.build_lm_input(llm_sd, torch.randint(0, 151936, (1, 48), generator=g), torch.zeros(1, 0, dtype=torch.long),
                                torch.zeros(1, 0, dtype=torch.long))
        _, cache0 = OLLM.qwen2_forward(llm_sd, cfg, x, None)

        def lm16():
            cache = cache0
            for i in range(16):
                y, cache = OLLM.qwen2_forward(llm_sd, cfg, llm_sd["speech_embedding.weight"][i].reshape(1, 1, -1), cache)
                torch.nn.functional.linear(y[:, -1], llm_sd["llm_decoder.weight"], llm_sd["llm_decoder.bias"]).log_softmax(-1)

        t_lm = [t / 16 for t in reps(lm16)]
        print(f"[cpu_baseline] LM {t_lm[0] * 1e3:.1f} ms/token (median {t_lm[1] * 1e3:.1f})", file=sys.stderr, flush=True)
        tok = torch.randint(0, 6561, (1, 250), generator=g)
        emb = torch.randn(1, 192, generator=g)
        t_flow = reps(lambda: OFLOW.flow_inference(flow_sd, tok, torch.zeros(1, 0, dtype=torch.long), torch.zeros(1, 0, 80), emb))
        print(f"[cpu_baseline] flow {t_flow[0]:.2f} s per 10 s utterance (median {t_flow[1]:.2f})", file=sys.stderr, flush=True)
        zl = torch.randn(1, 80, 500, generator=g)
        t_dac = reps(lambda: ODAC.decode(dac_sd, zl, [5, 4, 4, 3, 2]))
    tot_min = 250 * t_lm[0] + t_flow[0] + t_dac[0]
    tot_med = 250 * t_lm[1] + t_flow[1] + t_dac[1]
    return {"value": round(10.0 / tot_min, 4), "unit": "audio_s/s", "cores": cores, "kind": "port",
            "sample": "oracle, per stage 1 warm-up + 3 reps (min; median in value_median): 16 LM decode steps at ctx 50, flow on 250 "
                      "tokens (10 Euler steps, CFG), DAC on 500 frames; scaled to one 10 s utterance (250 tokens)",
            "value_median": round(10.0 / tot_med, 4), "rtf": round(tot_min / 10.0, 3),
            "llm_s_per_token": [round(v, 4) for v in t_lm], "flow_s_per_10s": [round(v, 3) for v in t_flow],
            "dac_s_per_10s": [round(v, 3) for v in t_dac]}


def _time_steps(fn, build, warmup, steps):
    """ms per call of fn: `build` untimed passes (eager pass + graph capture), `warmup`, then `steps` timed."""
    for _ in range(build + warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


ATTN = "bf16"
GROUP_FAN = None
FLOW_PRIO = None


def make_engine(dt, device, max_batch, max_ctx):
    """The engine under test (full-size synthetic weights).  A function of its own so that the 2-rank CPU rehearsal of
    this script's launch / shard / barrier / gather contract can substitute a stub (tests/test_cpu_host.py)."""
    from mmx.pipeline import TtsEngine
    if GROUP_FAN is not None:
        TtsEngine.group_fan = GROUP_FAN
    if FLOW_PRIO is not None:
        TtsEngine.flow_priority = FLOW_PRIO
    return TtsEngine(*build_weights(0), dtype=dt, device=device, max_batch=max_batch, max_ctx=max_ctx, attn=ATTN, wplanes="auto")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="x", choices=["bf16", "f32", "x"],
                    help="x (default) = the split build, the build that meets the north-star parity: bf16 weight stream and bf16 MFMA products, "
                         "fp32 activations split into bf16 terms inside the products; bf16 = the speed build (ids diverge from the CPU path)")
    ap.add_argument("--lm-prefetch", type=int, default=None, help="tuning: workgroups of the per-layer weight prefetch on a side stream of the decode step (LlmEngine.prefetch; 0 = off)")
    ap.add_argument("--lm-planes", default=None, choices=["f16x2", "bf16x3"], help="split build: plane format of the LM decode step (LlmEngine.lm_planes)")
    ap.add_argument("--checkpoint", default="bf16", choices=["bf16", "fp32"],
                    help="kind of the synthetic checkpoint (mmx/synth.py): bf16 = bf16-representable weights; fp32 = general fp32 weights and "
                         "trained-like weight norms, what the reference's loaders hand over - the split build then carries weight planes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the live kernel measurements after the timed region (profiler passes over the step itself)")
    ap.add_argument("--flow-bm-min", type=int, default=0, help="tuning: smallest tile height of the fused flow kernels")
    ap.add_argument("--flow-bm", type=int, default=0, help="cap the fused flow kernels' tile height (tuning: 32 leaves registers for co-resident decode waves)")
    ap.add_argument("--attn", default="bf16", choices=["bf16", "fp8"], help="fp8: estimator attention on the fp8 MFMA (config 5)")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-utterance and fp32-build extra measurements")
    ap.add_argument("--workload", default="batch", choices=["batch", "single", "longform"])
    ap.add_argument("--per-gpu", type=int, default=32)
    ap.add_argument("--flow-group", default="8", help="utterances per batched flow ODE solve (a list gives a ramp: k-th group)")
    ap.add_argument("--sched-adapt", action="store_true", help="tts_batch's cost model follows its own measurements (TtsEngine.sched_adapt)")
    ap.add_argument("--flash-form", type=int, default=None, help="lab: mmx_attn_flash_xs form of the flow groups beside the decode loop (0 chosen per launch, 1 = 128-query workgroups [default], 2 = 256-query, 3 = 4-wave 64-query)")
    ap.add_argument("--flow-priority", type=int, default=None, help="tuning: HIP stream priority of the flow workers' streams (TtsEngine.flow_priority)")
    ap.add_argument("--lm-cfg", default="", help="tuning: LlmEngine.v2_cfg overrides (output tiles per workgroup, k slices), e.g. gu=2,1:down=2,4")
    ap.add_argument("--group-fan", type=int, default=None, help="auxiliary streams per flow group for its per-utterance stages (TtsEngine.group_fan)")
    ap.add_argument("--pad-ratio", type=float, default=2.0, help="max length ratio inside one flow group")
    ap.add_argument("--flow-workers", type=int, default=2, help="host threads / streams solving flow groups concurrently")
    ap.add_argument("--poll-every", type=int, default=8, help="decode steps between two polls of the finished flags")
    ap.add_argument("--hold-steps", type=int, default=60, help="decode steps a finished utterance waits for a fuller flow group")
    ap.add_argument("--queue-batches", type=int, default=1, help="utterances per step = this many batches of --per-gpu, through the same decode slots (continuous batching: a freed slot admits the next utterance)")
    ap.add_argument("--gqa-min-batch", type=int, default=None, help="decode batches of at least this size use the GQA-shared attention kernel")
    ap.add_argument("--no-polite", action="store_true", help="flow groups beside the decode loop use the fastest tiling instead of the 64-row one")
    ap.add_argument("--tail-active", type=int, default=0, help="with at most this many sequences still decoding, finished utterances go to an idle flow worker at once")
    ap.add_argument("--no-overlap", action="store_true", help="run LM decode and flow/DAC back to back (one stream)")
    a = ap.parse_args()
    global ATTN, GROUP_FAN, FLOW_PRIO, CKPT_KIND
    ATTN, GROUP_FAN, FLOW_PRIO, CKPT_KIND = a.attn, a.group_fan, a.flow_priority, a.checkpoint
    if a.lm_planes:
        from mmx.llm import LlmEngine
        LlmEngine.lm_planes = a.lm_planes
    if a.lm_prefetch is not None:
        from mmx.llm import LlmEngine
        LlmEngine.prefetch = a.lm_prefetch
    if a.sched_adapt:
        from mmx.pipeline import TtsEngine
        TtsEngine.sched_adapt = True
    if a.flash_form is not None:
        from mmx.flow import FlowEngine
        FlowEngine.polite_flash_form_default = a.flash_form
    if a.lm_cfg:
        from mmx.llm import LlmEngine
        LlmEngine.v2_cfg = dict(LlmEngine.v2_cfg, **{kv.split("=")[0]: tuple(int(v) for v in kv.split("=")[1].split(",")) for kv in a.lm_cfg.split(":")})
    if a.gqa_min_batch is not None:
        from mmx.llm import LlmEngine
        LlmEngine.gqa_min_batch = a.gqa_min_batch
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    host = {"cores": None, "pinned": False}
    if world > 1:                                          # N ranks share one host: no rank may take every core for torch's CPU pools
        try:
            cpus = sorted(os.sched_getaffinity(0))
        except AttributeError:
            cpus = list(range(os.cpu_count() or world))
        share = max(1, len(cpus) // world)
        torch.set_num_threads(share)
        # every rank runs a decode thread and two flow workers: pin them to this rank's own share of the cores (contiguous, in
        # local-rank order) so that eight ranks do not migrate over each other's cores
        if len(cpus) >= 3 * world and hasattr(os, "sched_setaffinity") and not os.environ.get("MMX_NO_PIN"):
            mine_c = cpus[local * share:(local + 1) * share]
            try:
                os.sched_setaffinity(0, mine_c)
                host["pinned"] = True
            except OSError:
                pass
        host["cores"] = share
    # MMX_BENCH_REHEARSE=1: multi-rank rehearsal on a ONE-GPU box - every rank uses cuda:0, collectives run on gloo
    # over host copies.  Exercises the launch contract, sharding, barriers and the gather; never use it for numbers.
    rehearse = bool(os.environ.get("MMX_BENCH_REHEARSE")) and world > 1
    have_gpu = torch.cuda.is_available()
    if rehearse:
        local = 0
    if have_gpu:
        torch.cuda.set_device(local)
    dev = f"cuda:{local}" if have_gpu else "cpu"         # "cpu" only in the rehearsal test (tests/test_cpu_host.py stubs the engine)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    cdev = "cpu" if rehearse else "cuda"
    from mmx.pipeline import TOKEN_RATE, SAMPLE_RATE
    from mmx.dist import gather_audio, shard_utterances
    dt = {"bf16": 1, "f32": 0, "x": 2}[a.dtype]
    PER_GPU = 1 if a.workload in ("single", "longform") else a.per_gpu
    eng = make_engine(dt, dev, PER_GPU, 2048 if a.workload == "longform" else 640)
    if a.flow_bm:
        eng.flow.max_tile_rows = a.flow_bm
    if a.flow_bm_min:
        eng.flow.min_tile_rows = a.flow_bm_min
    emb = torch.randn(1, 192, generator=torch.Generator().manual_seed(1)).to(dev)
    if a.workload == "longform":
        # BASELINE config 5: one 60 s utterance per GPU, streaming (25-token hops, chunk-causal flow with the estimator /
        # encoder state cached between hops, captured decode step)
        lens_all = [1500] * world
    elif a.workload == "single":
        # BASELINE config 3 / SURVEY 8d.3: 48 random text ids, no prompt, exactly 250 decode steps (10 s of audio)
        lens_all = [250] * world
    else:
        # BASELINE config 4 / SURVEY 8d.4 (one rank's share): audio lengths U{2..20 s} = 50..500 tokens, seed 3,
        # PER_GPU utterances per GPU, dealt to the ranks by length (mmx.dist.shard_utterances)
        lens_all = torch.randint(50, 501, (PER_GPU * world,), generator=torch.Generator().manual_seed(3)).tolist() * a.queue_batches
    mine = shard_utterances(lens_all, world)[rank]
    lens = [lens_all[i] for i in mine]
    g = torch.Generator().manual_seed(2)
    all_text = [torch.randint(0, 151936, (1, 290 if a.workload == "longform" else 48), generator=g) for _ in range(len(lens_all))]
    texts = [all_text[i].to(dev) for i in mine]

    first_chunk_ms = []

    def step():
        if a.workload == "longform":
            if have_gpu:
                torch.cuda.synchronize()       # a new request on an idle device: the first-chunk latency is measured from here
            t_in = time.perf_counter()
            n = 0
            for k, w in enumerate(eng.tts_stream(texts[0], emb, seed=0, exact_steps=lens[0])):
                if k == 0:
                    if have_gpu:
                        torch.cuda.current_stream().synchronize()
                    first_chunk_ms.append((time.perf_counter() - t_in) * 1e3)
                n += w.shape[-1]
            return n
        wavs = eng.tts_batch(texts, [emb] * len(texts), seed=0, exact_steps=lens, group_size=[int(v) for v in str(a.flow_group).split(',')], overlap=not a.no_overlap, max_pad_ratio=a.pad_ratio, flow_workers=a.flow_workers, hold_steps=a.hold_steps, poll_every=a.poll_every, tail_active=a.tail_active, polite=not a.no_polite)
        if world > 1:                                              # the path's one exchange step (RCCL all-gather)
            gather_audio([w.cpu() for w in wavs] if rehearse else wavs, mine, len(lens_all))
        return sum(w.shape[-1] for w in wavs)

    def fence():
        if have_gpu:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            if have_gpu:
                torch.cuda.synchronize()

    # engine build, part of setup like weight packing: the first pass over a new shape runs eagerly and the second
    # records its hipGraphs (decode step per batch size, one Euler solve per flow group shape); W warm-up steps follow
    for _ in range(2):
        step()
    for _ in range(a.warmup):
        step()
    if os.environ.get("MMX_DUMP_MAPS"):
        # profiler runs: the process map at the start of the timed region (every library is loaded by now), so that the frames of
        # a fault report can be put back on their libraries (the rocprofv3 --pmc abort of round 3 could not be: no map was kept)
        with open("/proc/self/maps") as src, open(os.environ["MMX_DUMP_MAPS"], "w") as dst:
            dst.write(src.read())
    shape_log = []
    flows = getattr(eng, "_flows", None) or ([eng.flow] if hasattr(eng, "flow") else [])
    for fl in flows:                                       # the (n, T) of every flow group of the timed steps, for `roofline`
        fl.shape_log = shape_log
    fence()
    t0 = time.perf_counter()
    samples = 0
    for _ in range(a.steps):
        samples += step()
    fence()
    el = time.perf_counter() - t0
    for fl in flows:
        fl.shape_log = None
    if world > 1:
        t = torch.tensor([el], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        tot = torch.tensor([samples], device=cdev, dtype=torch.float64)
        dist.all_reduce(tot)
        samples = float(tot.item())
    audio_s = samples / SAMPLE_RATE
    if rank == 0:
        if a.workload == "longform":
            wl = ("BASELINE config 5: one 60 s utterance per GPU per step, streaming synthesis (290 text ids, 1500 AR decode "
                  "steps on the captured decode graph, 25-token hops: chunk-causal flow with cached K/V and conv state, so a hop "
                  f"solves only its 50 new frames, + DAC with exact left / right context), bf16 attention; first chunk after {sum(first_chunk_ms[-a.steps:]) / max(1, a.steps):.0f} ms")
        elif a.workload == "single":
            wl = ("BASELINE config 3: one 10 s utterance per GPU per step (48 text ids, 250 AR decode steps, flow 500 frames "
                  "x 10 Euler steps with CFG, DAC 240000 samples)")
        else:
            wl = (f"BASELINE config 4, one rank's share: {PER_GPU} utterances per GPU per step, lengths U{{2..20 s}} (50-500 "
                  "tokens, seed 3), batched AR decode, length-grouped batched flow (10 Euler steps, CFG), DAC decode, "
                  "all_gather of audio for N > 1")
        out = {"metric": "24 kHz audio-seconds generated per wall-second per node (CosyVoice2-0.5B-shaped LM + flow + DAC-VAE, end to end)",
               "value": round(audio_s / el, 3), "unit": "audio_s/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(el / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": {"x": "bf16", "bf16": "bf16", "f32": "f32"}[a.dtype] + ("+fp8 attention" if a.attn == "fp8" else ""), "data": "synthetic text ids, random-init weights (deterministic synth init)",
               "rtf": round(el / audio_s * world, 5),
               "config": {"workload": wl, "utterances_per_gpu": PER_GPU, "audio_s_per_step": round(audio_s / a.steps, 2),
                          "parallelism": f"dp{world} (replica per GPU, all_gather of audio)"}}
        out["collective"] = {"world_size": (dist.get_world_size() if world > 1 else 1), "backend": (dist.get_backend() if world > 1 else None)}
        lh = getattr(eng, "last_host", None)
        if lh:                                             # rank 0's last timed step: host issue time of the decode loop, when it ended, the whole call
            out["host"] = dict(host, **lh, lm_issue_us_per_decode_step=round(lh["lm_issue_ms"] * 1e3 / max(1, lh["decode_steps"]), 1),
                               sched_model=getattr(eng, "sched", None), sched_fit=getattr(eng, "sched_fit", None))
        BUILD = {2: "split (bf16 weight stream; fp32 activations carried as 2 (flow, DAC) / 3 (LM) bf16 terms inside the MFMA products)",
                 1: "bf16 (bf16 GEMM inputs, fp32 residual streams)", 0: "fp32 (f32-input MFMA)"}
        out["config"]["build"] = BUILD[dt]
        out["config"]["checkpoint"] = ("bf16-representable weights (exact in the bf16 / fp16 weight stream)" if a.checkpoint == "bf16" else
                                       "general fp32 weights, trained-like weight norms: the split build carries weight planes (2 fp16 planes in the LM "
                                       "decode step, 2 bf16 planes in the flow and the DAC)")
        out["north_star"] = {"requirement": "FSQ token ids bit-exact and waveform within 1e-3 abs of the CPU path on identical inputs (BASELINE.json)",
                             "met_by": "split" if dt == 2 else ("fp32" if dt == 0 else None),
                             "evidence": "tests/test_gpu_split.py: config 3 (250 / 250 ids), config 4 at this batch size (32 utterances, overlapped "
                                         "schedule) and config 5 (60 s stream) against the oracle; weights of the bf16 checkpoint kind (mmx/synth.py)"
                                         if dt == 2 else ("tests/test_gpu_pipeline.py" if dt == 0 else
                                                          "not met by this build (8-bit activations): see parity_build in this line")}

        def rooflines(e, d, log, nsteps):
            """roofline objects of engine `e` (build `d`): the whole LM decode step (`roofline`: the step's critical path), its
            largest projection, and the two flow kernels over the group shapes `log` of the timed steps."""
            r = {}
            if a.workload == "batch":
                r["roofline_lm_step"] = measure_lm_step(e)
            if d != 0:
                r["roofline_lm"] = measure_lm_kernel(e)
            r["roofline"] = r.get("roofline_lm_step") or r.get("roofline_lm")
            if d in (1, 2) and a.workload == "batch" and log:
                r["roofline_flow"] = measure_flow_kernel(e, log, nsteps)
                r["roofline_attn"] = measure_attn_kernel(e, log, nsteps)
            return r

        if world == 1 and not a.no_roofline:
            # `roofline`: the LM decode step as a whole - its projection family is the largest share of the step's GPU time
            # (profiles/r04_bench*_kernel_stats.csv) and the decode loop is the step's critical path; the other objects are the
            # largest single projection (roofline_lm) and the two flow kernels (roofline_flow / _attn)
            out.update(rooflines(eng, dt, shape_log, a.steps))
        if world == 1:
            if a.workload == "batch" and not a.no_extras:
                # extra keys, measured after the timed region
                from mmx.pipeline import TtsEngine
                # (0) continuous batching across batch boundaries: three of the step's batches (96 utterances) through
                # the same 32 decode slots in ONE call - a freed slot admits the next queued utterance (LlmEngine.admit),
                # so the decode batch stays full instead of draining to its longest member at the end of every batch
                QB = 3
                g3 = torch.Generator().manual_seed(5)
                texts_q = [torch.randint(0, 151936, (1, 48), generator=g3).to(dev) for _ in range(QB * len(lens))]
                lens_q = sorted(lens * QB, reverse=True)       # longest first, like shard_utterances deals them
                fq = lambda: eng.tts_batch(texts_q, [emb] * len(texts_q), seed=0, exact_steps=lens_q)
                msq = _time_steps(fq, build=2, warmup=0, steps=2)
                out["continuous_batching"] = {"utterances_per_call": len(lens_q), "decode_slots": PER_GPU, "ms_per_call": round(msq, 1),
                                              "value": round(sum(lens_q) / TOKEN_RATE / (msq / 1e3), 2), "unit": "audio_s/s",
                                              "note": f"{QB} batches of the step's workload queued into one call; not the headline "
                                                      "(a step there is one batch, finished before the next starts)"}
                # (0b) the north-star target's own workload: a batch of UNIFORM 10 s utterances (32 x 250 tokens).  All sequences
                # finish together, so nothing of the flow / DAC stage overlaps the decode loop inside one batch; with batches
                # queued the next batch's decode runs beside this batch's flow.
                uni = [250] * len(lens)
                fu = lambda: eng.tts_batch(texts, [emb] * len(texts), seed=0, exact_steps=uni)
                msu = _time_steps(fu, build=2, warmup=0, steps=2)
                uni_q = uni * QB
                fuq = lambda: eng.tts_batch(texts_q, [emb] * len(texts_q), seed=0, exact_steps=uni_q)
                msuq = _time_steps(fuq, build=2, warmup=0, steps=2)
                out["uniform_10s"] = {"workload": f"{len(uni)} utterances of exactly 250 tokens (10 s) per step", "ms_per_step": round(msu, 1),
                                      "value": round(sum(uni) / TOKEN_RATE / (msu / 1e3), 2), "unit": "audio_s/s",
                                      "rtf_per_utterance": round(msu / 1e3 / 10.0, 5),
                                      "queued": {"utterances_per_call": len(uni_q), "ms_per_call": round(msuq, 1),
                                                 "value": round(sum(uni_q) / TOKEN_RATE / (msuq / 1e3), 2)}}
                # (0c) zero-shot synthesis (the reference's main entry, cli/cosyvoice.py:92-104): the step's workload with a 3 s
                # prompt per utterance - 8 prompt text ids + 75 prompt speech tokens in front of the LM input, 75 prompt tokens
                # + 150 prompt latent frames through the flow (their frames are solved and dropped, flow.py:472-509)
                gz = torch.Generator().manual_seed(9)
                zs = dict(prompt_texts=[torch.randint(0, 151936, (1, 8), generator=gz).to(dev) for _ in lens],
                          llm_prompt_speech_tokens=[torch.randint(0, 6561, (1, 75), generator=gz).to(dev) for _ in lens],
                          flow_prompt_speech_tokens=[torch.randint(0, 6561, (1, 75), generator=gz).to(dev) for _ in lens],
                          prompt_speech_feats=[(torch.randn(1, 150, 80, generator=gz) * 0.5).to(dev) for _ in lens])
                fz = lambda: eng.tts_batch(texts, [emb] * len(texts), seed=0, exact_steps=lens, **zs)
                msz = _time_steps(fz, build=2, warmup=0, steps=2)
                out["zero_shot"] = {"workload": "the step's 32 utterances, each with a 3 s prompt (8 text ids + 75 speech tokens to the LM, 75 "
                                                "tokens + 150 latent frames to the flow)", "ms_per_step": round(msz, 1),
                                    "value": round(sum(lens) / TOKEN_RATE / (msz / 1e3), 2), "unit": "audio_s/s (generated audio only)"}
                eng.close()
                del eng
                torch.cuda.empty_cache()
                w3 = build_weights(0)
                # (1) BASELINE config 3 (one 10 s utterance) for the per-utterance RTF target (>= 10x real time), this build
                e1 = TtsEngine(*w3, dtype=dt, device=f"cuda:{local}", max_batch=1, max_ctx=640, wplanes="auto")
                ms1 = _time_steps(lambda: e1.tts(all_text[0].cuda(), emb, seed=0, exact_steps=250), build=2, warmup=1, steps=5)
                out["single_utterance"] = {"workload": "BASELINE config 3: 48 text ids, 250 AR decode steps, flow 500 frames x 10 Euler steps, "
                                           "DAC 240000 samples", "build": a.dtype, "ms": round(ms1, 2), "rtf": round(ms1 / 1e4, 5),
                                           "x_realtime": round(1e4 / ms1, 1)}
                e1.close()
                del e1
                # (2) the same config-4 share on the OTHER fast build, timed like the headline (build passes, --warmup, then
                # --steps steps) with its own roofline objects: `speed_build` (bf16: 8-bit activations, does not reproduce the
                # oracle's ids) when the headline is the split build, `parity_build` (split) when the headline is bf16
                d2 = {2: 1, 1: 2}.get(dt)
                if d2 is not None:
                    torch.cuda.empty_cache()
                    ef = TtsEngine(*w3, dtype=d2, device=f"cuda:{local}", max_batch=PER_GPU, max_ctx=640, wplanes="auto")
                    fn = lambda: ef.tts_batch(texts, [emb] * len(texts), seed=0, exact_steps=lens)
                    for _ in range(2 + a.warmup):
                        fn()
                    log2 = []
                    fl2 = getattr(ef, "_flows", None) or [ef.flow]
                    for fl in fl2:
                        fl.shape_log = log2
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(a.steps):
                        fn()
                    torch.cuda.synchronize()
                    msf = (time.perf_counter() - t1) / a.steps * 1e3
                    for fl in fl2:
                        fl.shape_log = None
                    a_s = sum(lens) / TOKEN_RATE
                    ob = {"build": BUILD[d2], "value": round(a_s / (msf / 1e3), 2), "unit": "audio_s/s", "ms_per_step": round(msf, 1),
                          "steps": a.steps, "warmup": a.warmup,
                          "meets_north_star": d2 == 2,
                          "note": ("the same workload on the bf16 build: faster, but its token ids diverge from the CPU path's "
                                   "(tests/test_gpu_pipeline.py::test_composed_pipeline_bf16_bound states its bound)") if d2 == 1 else
                                  ("the same workload on the split build, the build held to ids identical / waveform <= 1e-3 at this "
                                   "batch size by tests/test_gpu_split.py::test_config4_rank_share_full_size_split_vs_oracle")}
                    ob.update(rooflines(ef, d2, log2, a.steps))
                    out["speed_build" if d2 == 1 else "parity_build"] = ob
                    ef.close()
                    del ef
                del w3
            if not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
