"""oracle/philox.py — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Counter-based sampling noise shared bit-for-bit by the CPU oracle and the HIP sampler
(minimax-speech_amd/csrc/sampler.hip uses the same Philox4x32-10 constants and the same
counter layout), replacing the reference's torch CPU mt19937 stream, which a GPU cannot replay
(SURVEY.md §7 "Bit-exact FSQ token ids vs sampling").

The reference draws via torch.multinomial(p, 1), which on CPU is argmax(p / e), e ~ Exp(1) i.i.d.
per category (verified against torch in tests/test_oracle_sampling.py).  Here
    e_i = -log(u_i),  u_i = ((w_i >> 8) + 0.5) * 2^-24  in (0,1),
    w_i = word (i & 3) of Philox4x32-10(counter = (i >> 2, 2*trial + which, step, seq), key = (seed_lo, seed_hi))
with which = 0 for the nucleus draw and 1 for the full-vocabulary ("random_sampling") draw.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    """Vectorised over numpy uint32 arrays c0..c3; scalar keys."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def exp_noise(seed: int, seq: int, step: int, trial: int, which: int, n: int) -> np.ndarray:
    """float32 e[n] ~ Exp(1) for one multinomial draw (see module docstring)."""
    i = np.arange(n, dtype=np.uint32)
    w = philox4x32(i >> np.uint32(2), np.uint32(2 * trial + which), np.uint32(step), np.uint32(seq),
                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    w = np.stack(w, axis=0)[(i & np.uint32(3)).astype(np.int64), np.arange(n)]
    u = ((w >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
    return (-np.log(u.astype(np.float32))).astype(np.float32)
