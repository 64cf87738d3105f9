"""Why do the split build's token ids leave the oracle's at step ~1143 of the 60 s utterance (BASELINE config 5)?
Teacher-forced comparison of the LM's log-probs along the oracle's own id sequence: per step the largest |dlogp| over the
oracle's nucleus candidates (split build X3 and fp32 build against the CPU oracle), and the oracle's DECISION MARGINS at every
step - the relative gap between the two largest p_i / e_i of the multinomial race, and the distance of the running sum from
top_p at the nucleus cut.  A flip at a step whose margin is of the order of the log-prob error is a near-tie, not a defect.

    python tools/long_ctx_diag.py [steps]        (needs the GPU; ~3 min of CPU oracle)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "minimax-speech_amd"))
from mmx import shapes, synth  # noqa: E402
from mmx.llm import LlmEngine  # noqa: E402
from oracle import llm as OLLM  # noqa: E402
from oracle.philox import exp_noise  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1160
SEED = 1
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
sd = synth.synth_state_dict(shapes.llm_manifest(), 0)
text = torch.randint(0, 151936, (1, 290), generator=torch.Generator().manual_seed(6))
z = torch.zeros(1, 0, dtype=torch.long)
rec = []
with torch.no_grad():
    toks = OLLM.lm_inference(sd, OLLM.QwenCfg(), text, z, z, seed=SEED, seq=0, max_steps=N, ignore_eos_always=True, record=rec)
print(f"oracle: {len(rec)} steps, {len(toks)} accepted ids", flush=True)
# the sampled id of every step (ids above EOS are skipped, not accepted): re-run the sampler on the recorded log-probs
out, sampled, margins, cutm, fallback = [], [], [], [], []
for i, lp in enumerate(rec):
    prob, idx = OLLM.nucleus_candidates(lp)
    e = torch.from_numpy(exp_noise(SEED, 0, i, 0, 0, prob.numel()))
    r = (prob / e).sort(descending=True)
    m0 = float((r.values[0] - r.values[1]) / r.values[0]) if prob.numel() > 1 else 1.0
    # the repetition-aware fallback (common.py:113-115): a full-vocabulary race decides the step; its margin is what counts there
    nuc_top = int(idx[r.indices[0]])
    fb = sum(1 for t in out[-10:] if t == nuc_top) >= 1
    fallback.append(fb)
    if fb:
        p_all = lp.softmax(0)
        rr = (p_all / torch.from_numpy(exp_noise(SEED, 0, i, 0, 1, p_all.numel()))).sort(descending=True).values
        m0 = min(m0, float((rr[0] - rr[1]) / rr[0]))
    margins.append(m0)
    sv = lp.softmax(0).sort(descending=True, stable=True).values
    cum = torch.cumsum(sv[:26], 0)
    k = prob.numel()
    cutm.append(min(abs(float(cum[k - 1]) - 0.8), abs(float(cum[k - 2]) - 0.8) if k > 1 else 1.0))
    top = OLLM.sampling_ids_e(lp, out, lambda kk, i=i: OLLM.philox_noise(SEED, 0, i, kk), ignore_eos=True, eos=6561)
    sampled.append(top)
    if top < 6561:
        out.append(top)
assert out == toks
forced = torch.tensor(sampled).reshape(1, -1)
res = {}
for name, dt in (("split X3", 3), ("fp32", 0)):
    eng = LlmEngine(sd, dtype=dt, max_batch=1, max_ctx=2048, use_graphs=False)
    x = eng.build_lm_input(text.cuda(), z.cuda(), z.cuda())
    eng.start([x], [N], [N], seed=SEED, forced=forced.cuda(), want_logp=True)
    lps, drawn = [eng.logp[0].cpu().clone()], None
    for i in range(1, len(rec)):
        eng.step()
        lps.append(eng.logp[0].cpu().clone())
    drawn = eng.sampled[0, :len(rec)].tolist()
    d, dall = [], []
    for i, (a, b) in enumerate(zip(lps, rec)):
        _, idx = OLLM.nucleus_candidates(b)
        d.append(float((a[idx] - b[idx]).abs().max()))
        dall.append(float((a - b).abs().max()))
    res[name] = (d, drawn, dall, lps)
    eng.close()
    del eng
    torch.cuda.empty_cache()
for name, (d, drawn, dall, lps) in res.items():
    t = torch.tensor(d)
    print(f"\n{name}: max |dlogp| over the oracle's nucleus candidates, teacher forced")
    for lo in range(0, len(d), 200):
        seg = t[lo:lo + 200]
        print(f"   steps {lo:4d}..{lo + len(seg) - 1:4d} (context {292 + lo}..): mean {seg.mean():.2e}  max {seg.max():.2e}")
    flips = [i for i, (a, b) in enumerate(zip(drawn, sampled)) if a != b]
    print(f"   draws that differ from the oracle's: {len(flips)} of {len(d)} at steps {flips[:10]}")
    print(f"   max |dlogp| over ALL {len(rec[0])} ids, all steps: {max(dall):.2e}")
    for i in flips[:10]:
        print(f"      step {i}: oracle drew {sampled[i]}, this build {drawn[i]}; fallback (full-vocabulary race) {fallback[i]}; smallest race margin "
              f"{margins[i]:.2e}, nucleus-cut margin {cutm[i]:.2e}, |dlogp| candidates {d[i]:.2e} / all ids {dall[i]:.2e}; "
              f"logp of the two ids: oracle {float(rec[i][sampled[i]]):.6f} {float(rec[i][drawn[i]]):.6f}, build {float(lps[i][sampled[i]]):.6f} {float(lps[i][drawn[i]]):.6f}")
m = torch.tensor(margins)
print(f"\nfallback steps: {sum(fallback)} of {len(fallback)}")
print(f"oracle decision margins over {len(m)} steps: race margin min {m.min():.2e} (step {int(m.argmin())}), "
      f"steps with margin < 1e-4: {int((m < 1e-4).sum())}, < 1e-3: {int((m < 1e-3).sum())}; nucleus-cut margin min {min(cutm):.2e}")
