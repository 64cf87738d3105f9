"""bench.py — headline benchmark of the TTS hot path (BASELINE.json metric: 24 kHz audio-seconds per wall-second
per node, + RTF), CosyVoice2-0.5B-shaped models, random-init weights, synthetic inputs.

One "step" = one full pass of the path over the rank's batch: AR LM decode (text ids -> FSQ tokens), flow-matching
decoder (tokens -> latents, 10 Euler steps, CFG), DAC-VAE decoder (latents -> 24 kHz waveform) and, for N > 1, the
RCCL all-gather of the generated audio.  Inputs are resident on the device before the timed region.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
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


def build_weights(seed=0):
    from mmx import shapes, synth
    return (synth.synth_state_dict(shapes.llm_manifest(), seed), synth.synth_state_dict(shapes.flow_manifest(), seed),
            synth.synth_state_dict(shapes.dac_decoder_manifest(80), seed))


def measure_dominant_kernel(eng, iters=240):
    """Roofline of the dominant kernel of the step: the weight-streaming GEMM of the LM decode (skinny_gemm_kernel),
    gate/up projection + SwiGLU instance (the largest of the 4 per layer), at the batch size of the workload.
    Timed live with HIP events on the stream the kernel is launched on, rotating over the 24 layers' weights so
    the 256 MiB Infinity Cache cannot serve the stream.  Algorithmic bytes per launch (bf16 weights
    are the traffic of a decode step) = the packed bf16 weight matrix 2 x 4864 x 896 x 2 B + the activation rows in
    and out.  `traffic` = HBM bytes per launch from rocprofv3 PMC (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE),
    collected in separate passes with tools/pmc_skinny.py and committed as profiles/r01_pmc_skinny.json."""
    from mmx import ops
    llm = eng.llm
    B, H, I = llm.B, llm.H, llm.I
    pk = llm.packed                              # the activation layout the decode step uses at this batch size
    x = torch.randn(B, H, device=llm.dev).to(llm.tdt)
    if pk:
        x = ops.pack_act(x, llm.dtype)
    act = torch.empty(ops.packed_rows(B), I, dtype=llm.tdt, device=llm.dev)
    esz = 2 if llm.dtype == 1 else 4
    nbytes = 2 * I * H * esz + B * H * esz + B * I * esz
    s = torch.cuda.current_stream()
    run = lambda l: ops.skinny_gemm(x, llm.layers[l]["wgu"], B=B, K=H, N=I, dtype=llm.dtype, rs=True, eps=llm.eps, epi=1, out_act=act,
                                    x_packed=pk, out_packed=pk)
    for l in range(llm.n_layers):
        run(l)
    g = torch.cuda.CUDAGraph()                  # same launch mechanism as the decode step (hipGraph replay)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for i in range(iters):
            run(i % llm.n_layers)
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    g.replay()
    e1.record(torch.cuda.current_stream())
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    achieved = nbytes / (us * 1e-6) / 1e9
    traffic = None
    pj = os.path.join(ROOT, "profiles", "r01_pmc_skinny.json")
    if os.path.exists(pj) and B in (1, 32):
        traffic = json.load(open(pj)).get("hbm_bytes_per_launch" if B == 1 else "batch32_hbm_bytes_per_launch")
    return {"bound": "hbm", "kernel": f"skinny_gemm_kernel (LM gate/up + SwiGLU, K=896, N=2x4864, batch {B})",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic, "bytes_per_launch": nbytes, "us_per_launch": round(us, 3)}


def cpu_baseline():
    """The CPU oracle (port of the reference algorithm, oracle/*.py) timed on this host's cores on a bounded
    sample of the same workload: 16 LM decode steps (after a 50-row prefill), the flow on 25 tokens (50 frames,
    10 Euler steps) and the DAC decoder on 50 frames; per-stage cost is scaled to a 10 s utterance."""
    from oracle import dac as ODAC, flow as OFLOW, llm as OLLM
    from mmx import shapes, synth
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
    with torch.no_grad():
        cfg = OLLM.QwenCfg()
        x = OLLM.build_lm_input(llm_sd, torch.randint(0, 151936, (1, 48), generator=g), torch.zeros(1, 0, dtype=torch.long),
                                torch.zeros(1, 0, dtype=torch.long))
        y, cache = OLLM.qwen2_forward(llm_sd, cfg, x, None)
        t0 = time.time()
        for i in range(16):
            y, cache = OLLM.qwen2_forward(llm_sd, cfg, llm_sd["speech_embedding.weight"][i].reshape(1, 1, -1), cache)
            torch.nn.functional.linear(y[:, -1], llm_sd["llm_decoder.weight"], llm_sd["llm_decoder.bias"]).log_softmax(-1)
        t_tok = (time.time() - t0) / 16
        print(f"[cpu_baseline] LM {t_tok * 1e3:.1f} ms/token", file=sys.stderr, flush=True)
        tok = torch.randint(0, 6561, (1, 25), generator=g)
        t0 = time.time()
        OFLOW.flow_inference(flow_sd, tok, torch.zeros(1, 0, dtype=torch.long), torch.zeros(1, 0, 80), torch.randn(1, 192, generator=g))
        t_flow = time.time() - t0                              # 1 s of audio
        print(f"[cpu_baseline] flow {t_flow:.2f} s per audio-second", file=sys.stderr, flush=True)
        t0 = time.time()
        ODAC.decode(dac_sd, torch.randn(1, 80, 50, generator=g), [5, 4, 4, 3, 2])
        t_dac = time.time() - t0                               # 1 s of audio
    per_audio_s = 25 * t_tok + t_flow + t_dac
    return {"value": round(1.0 / per_audio_s, 4), "unit": "audio_s/s", "cores": cores, "kind": "port",
            "sample": "oracle: 16 LM decode steps + flow on 25 tokens (10 Euler steps) + DAC on 50 frames, scaled per audio-second",
            "llm_s_per_token": round(t_tok, 4), "flow_s_per_audio_s": round(t_flow, 3), "dac_s_per_audio_s": round(t_dac, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="batch", choices=["batch", "single", "longform"])
    ap.add_argument("--per-gpu", type=int, default=32)
    ap.add_argument("--flow-group", default="2,2,4,8", help="utterances per batched flow ODE solve (ramp: k-th group)")
    ap.add_argument("--pad-ratio", type=float, default=2.0, help="max length ratio inside one flow group")
    ap.add_argument("--flow-workers", type=int, default=2, help="host threads / streams solving flow groups concurrently")
    ap.add_argument("--poll-every", type=int, default=8, help="decode steps between two polls of the finished flags")
    ap.add_argument("--hold-steps", type=int, default=40, help="decode steps a finished utterance waits for a fuller flow group")
    ap.add_argument("--no-overlap", action="store_true", help="run LM decode and flow/DAC back to back (one stream)")
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    # MMX_BENCH_REHEARSE=1: multi-rank rehearsal on a ONE-GPU box - every rank uses cuda:0, collectives run on gloo
    # over host copies.  Exercises the launch contract, sharding, barriers and the gather; never use it for numbers.
    rehearse = bool(os.environ.get("MMX_BENCH_REHEARSE")) and world > 1
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    cdev = "cpu" if rehearse else "cuda"
    from mmx.pipeline import TtsEngine, TOKEN_RATE, SAMPLE_RATE
    from mmx.dist import gather_audio, shard_utterances
    dt = 1 if a.dtype == "bf16" else 0
    llm_sd, flow_sd, dac_sd = build_weights(0)
    PER_GPU = 1 if a.workload in ("single", "longform") else a.per_gpu
    eng = TtsEngine(llm_sd, flow_sd, dac_sd, dtype=dt, device=f"cuda:{local}", max_batch=PER_GPU,
                    max_ctx=2048 if a.workload == "longform" else 640)
    del llm_sd, flow_sd, dac_sd
    emb = torch.randn(1, 192, generator=torch.Generator().manual_seed(1)).cuda()
    if a.workload == "longform":
        # BASELINE config 5: one 60 s utterance per GPU, streaming (25-token hops, chunk-causal flow over all tokens so
        # far at every hop, captured decode step); bf16 attention (the fp8 MFMA variant is not built)
        lens_all = [1500] * world
    elif a.workload == "single":
        # BASELINE config 3 / SURVEY 8d.3: 48 random text ids, no prompt, exactly 250 decode steps (10 s of audio)
        lens_all = [250] * world
    else:
        # BASELINE config 4 / SURVEY 8d.4 (one rank's share): audio lengths U{2..20 s} = 50..500 tokens, seed 3,
        # PER_GPU utterances per GPU, dealt to the ranks by length (mmx.dist.shard_utterances)
        lens_all = torch.randint(50, 501, (PER_GPU * world,), generator=torch.Generator().manual_seed(3)).tolist()
    mine = shard_utterances(lens_all, world)[rank]
    lens = [lens_all[i] for i in mine]
    g = torch.Generator().manual_seed(2)
    all_text = [torch.randint(0, 151936, (1, 290 if a.workload == "longform" else 48), generator=g) for _ in range(len(lens_all))]
    texts = [all_text[i].cuda() for i in mine]
    max_samples = 2 * max(lens_all) * eng.hop

    first_chunk_ms = []

    def step():
        if a.workload == "longform":
            t_in = time.perf_counter()
            n = 0
            for k, w in enumerate(eng.tts_stream(texts[0], emb, seed=0, exact_steps=lens[0])):
                if k == 0:
                    torch.cuda.current_stream().synchronize()
                    first_chunk_ms.append((time.perf_counter() - t_in) * 1e3)
                n += w.shape[-1]
            return n
        wavs = eng.tts_batch(texts, [emb] * len(texts), seed=0, exact_steps=lens, group_size=[int(v) for v in str(a.flow_group).split(',')], overlap=not a.no_overlap, max_pad_ratio=a.pad_ratio, flow_workers=a.flow_workers, hold_steps=a.hold_steps, poll_every=a.poll_every)
        if world > 1:                                              # the path's one exchange step (RCCL all-gather)
            gather_audio([w.cpu() for w in wavs] if rehearse else wavs, mine, len(lens_all), max_samples)
        return sum(w.shape[-1] for w in wavs)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # engine build, part of setup like weight packing: the first pass over a new shape runs eagerly and the second
    # records its hipGraphs (decode step per batch size, one Euler solve per flow group shape); W warm-up steps follow
    for _ in range(2):
        step()
    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    samples = 0
    for _ in range(a.steps):
        samples += step()
    fence()
    el = time.perf_counter() - t0
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
                  "steps on the captured decode graph, 25-token hops: chunk-causal flow over all tokens so far + DAC of the new "
                  f"frames), bf16 attention; first chunk after {sum(first_chunk_ms[-a.steps:]) / max(1, a.steps):.0f} ms")
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
               "dtype": a.dtype, "data": "synthetic text ids, random-init weights (deterministic synth init)",
               "rtf": round(el / audio_s * world, 5),
               "config": {"workload": wl, "utterances_per_gpu": PER_GPU, "audio_s_per_step": round(audio_s / a.steps, 2),
                          "parallelism": f"dp{world} (replica per GPU, all_gather of audio)"}}
        if world == 1:
            out["roofline"] = measure_dominant_kernel(eng)
            if not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
