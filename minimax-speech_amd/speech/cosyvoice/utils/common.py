"""Drop-in for speech/cosyvoice/utils/common.py (the functions the inference hot path uses).

The product path samples INSIDE the decode graph (csrc/sampler.hip).  These host-side functions keep the
reference's signatures so that `!name:cosyvoice.utils.common.ras_sampling` in config.yaml:46-50 still binds
(Qwen2LM reads top_p/top_k/win_size/tau_r from the partial) and callers that sample on their own keep working."""
import random

import numpy as np
import torch

IGNORE_ID = -1


def nucleus_sampling(weighted_scores, top_p=0.8, top_k=25):
    """common.py:119-134: stable descending sort, keep while cum < top_p and n < top_k, multinomial."""
    sv, si = weighted_scores.softmax(dim=0).sort(descending=True, stable=True)
    cum = torch.cumsum(sv, 0)
    # element i is kept iff the running sum BEFORE it is < top_p and i < top_k
    before = cum - sv
    n = int(((before < top_p) & (torch.arange(sv.numel(), device=sv.device) < top_k)).sum())
    n = max(n, 1)
    return si[:n][sv[:n].multinomial(1, replacement=True)]


def random_sampling(weighted_scores, decoded_tokens, sampling):
    return weighted_scores.softmax(dim=0).multinomial(1, replacement=True)


def ras_sampling(weighted_scores, decoded_tokens, sampling, top_p=0.8, top_k=25, win_size=10, tau_r=0.1):
    """Repetition Aware Sampling (common.py:111-116)."""
    top_ids = nucleus_sampling(weighted_scores, top_p=top_p, top_k=top_k)
    rep = (torch.tensor(decoded_tokens[-win_size:], device=weighted_scores.device) == top_ids).sum().item()
    if rep >= win_size * tau_r:
        top_ids = random_sampling(weighted_scores, decoded_tokens, sampling)
    return top_ids


def fade_in_out(fade_in_mel, fade_out_mel, window):
    """common.py:142-150: cross-fade the head of `fade_in_mel` with the tail of `fade_out_mel`."""
    n = int(window.shape[0] / 2)
    out = fade_in_mel.clone()
    w = torch.as_tensor(window, dtype=out.dtype, device=out.device)
    out[..., :n] = out[..., :n] * w[:n] + fade_out_mel[..., -n:].to(out.device) * w[n:]
    return out


def set_all_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def mask_to_bias(mask: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    assert mask.dtype == torch.bool
    assert dtype in [torch.float32, torch.bfloat16, torch.float16]
    return (1.0 - mask.to(dtype)) * -1.0e+10
