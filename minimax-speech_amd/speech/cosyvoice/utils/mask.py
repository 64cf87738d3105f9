"""Drop-in for speech/cosyvoice/utils/mask.py (:127-158 subsequent_chunk_mask, :161-236 add_optional_chunk_mask,
:239-265 make_pad_mask).  Index plumbing only; the HIP attention kernels take the same visibility rule as
(key mask, chunk size) instead of a materialised [B,T,T] tensor."""
import torch


def make_pad_mask(lengths: torch.Tensor, max_len: int = 0) -> torch.Tensor:
    max_len = max_len if max_len > 0 else int(lengths.max().item())
    rng = torch.arange(0, max_len, dtype=torch.int64, device=lengths.device)
    return rng.unsqueeze(0) >= lengths.unsqueeze(-1)


def subsequent_chunk_mask(size: int, chunk_size: int, num_left_chunks: int = -1,
                          device: torch.device = torch.device("cpu")) -> torch.Tensor:
    pos = torch.arange(size, device=device)
    limit = (torch.div(pos, chunk_size, rounding_mode="trunc") + 1) * chunk_size
    return pos.unsqueeze(0) < limit.unsqueeze(1)


def add_optional_chunk_mask(xs, masks, use_dynamic_chunk, use_dynamic_left_chunk, decoding_chunk_size,
                            static_chunk_size, num_decoding_left_chunks, enable_full_context=True):
    if use_dynamic_chunk:
        raise NotImplementedError("dynamic chunk training masks are outside the inference hot path")
    if static_chunk_size > 0:
        chunk_masks = masks & subsequent_chunk_mask(xs.size(1), static_chunk_size, num_decoding_left_chunks, xs.device).unsqueeze(0)
    else:
        chunk_masks = masks
    dead = chunk_masks.sum(dim=-1) == 0
    if dead.any():
        chunk_masks = chunk_masks.clone()
        chunk_masks[dead] = True
    return chunk_masks
