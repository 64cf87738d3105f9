"""Multi-GPU data parallelism for the TTS hot path: utterances are independent, every rank holds
a full replica, the only exchange is an all-gather of the generated audio (RCCL over xGMI on the GPU box;
the same code runs on gloo/CPU in tests).  No other collective exists on this path."""
from typing import List, Optional, Sequence, Tuple

import torch


def shard_utterances(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Length-balanced partition: sort by expected length (longest first) and deal round-robin in a snake order,
    so every rank gets the same number of utterances (+-1) and nearly the same number of frames.
    Returns, per rank, the indices (into `lengths`) it owns."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    shards: List[List[int]] = [[] for _ in range(world)]
    for pos, idx in enumerate(order):
        rnd, k = divmod(pos, world)
        shards[k if rnd % 2 == 0 else world - 1 - k].append(idx)
    return shards


_gather_bufs = {}        # (rows, max_samples, device) -> all_buf: allocated once per shape, reused by every call
_meta_bufs = {}          # (rows, device) -> all_meta


def gather_audio(wavs: List[torch.Tensor], owned: List[int], n_total: int, max_samples: Optional[int] = None,
                 group=None, quantum: int = 24000) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """All ranks end up with every utterance.  Returns (buf, lengths, row_of): utterance i (global index) is
    buf[row_of[i], :lengths[i]]; buf is THE receive buffer of the all-gather, [world * per_rank, slot] fp32 zero
    padded, in rank-major slot order — no second copy in global order is made.
    Two collectives: (index, sample count) pairs first, then the padded audio.  max_samples=None sizes a slot
    by what the FIRST collective reports — the longest utterance of this call, rounded up to `quantum` samples (1 s: few distinct
    buffer shapes) — instead of by the longest utterance the job could produce (8 x 32 slots of 20 s were 491 MB per rank and
    call for ~265 MB of audio); a caller that needs one fixed shape (capture) passes max_samples."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = wavs[0].device if wavs else torch.device("cpu")
    per = (n_total + world - 1) // world
    assert len(wavs) <= per
    mkey = (world * per, str(dev))
    if mkey not in _meta_bufs:
        _meta_bufs.clear()
        _meta_bufs[mkey] = torch.empty((world * per, 2), dtype=torch.int64, device=dev)
    all_meta = _meta_bufs[mkey]
    meta = all_meta[rank * per:(rank + 1) * per]
    meta_h = torch.full((per, 2), -1, dtype=torch.int64)        # (global index, n samples), one host->device copy
    for slot, (w, gi) in enumerate(zip(wavs, owned)):
        meta_h[slot, 0], meta_h[slot, 1] = gi, w.numel()
    if world == 1:
        all_meta.fill_(-1)
    meta.copy_(meta_h)
    if world > 1:
        dist.all_gather_into_tensor(all_meta, meta.clone(), group=group)
    if max_samples is None:
        longest = int(all_meta[:, 1].max().item())              # every rank computes the same slot size from the same counts
        max_samples = max(quantum, -(-longest // quantum) * quantum)
    key = (world * per, int(max_samples), str(dev))
    if key not in _gather_bufs:
        _gather_bufs.clear()                                   # one live shape at a time
        _gather_bufs[key] = torch.zeros(world * per, max_samples, dtype=torch.float32, device=dev)
    all_buf = _gather_bufs[key]
    buf = all_buf[rank * per:(rank + 1) * per]                  # this rank's slots, in place
    for slot, w in enumerate(wavs):
        n = w.numel()
        assert n <= max_samples
        buf[slot, :n] = w.reshape(-1)
        buf[slot, n:].zero_()
    for slot in range(len(wavs), per):
        buf[slot].zero_()
    if world > 1:
        dist.all_gather_into_tensor(all_buf, buf, group=group)      # in-place form: the input is this rank's slice of the output
    lens = torch.zeros(n_total, dtype=torch.int64, device=dev)
    row_of = torch.full((n_total,), -1, dtype=torch.int64, device=dev)
    valid = all_meta[:, 0] >= 0
    rows = torch.nonzero(valid).reshape(-1)
    row_of[all_meta[rows, 0]] = rows
    lens[all_meta[rows, 0]] = all_meta[rows, 1]
    return all_buf, lens, row_of
