"""oracle/ — TEST INFRASTRUCTURE ONLY.

CPU (torch fp32) restatement of the reference's algorithm for the TTS inference hot path
(SURVEY.md §8a rows a1–a11).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this package, and only as the checker — the
product path (`minimax-speech_amd/`) never does and fails loudly without its HIP library.

Pinning: the reference ships NO golden vectors or known-answer tests for this path
(SURVEY.md §4).  The oracle is pinned instead by outputs of the reference's own Python
classes, run in the build container through `oracle/ref_shims.py` and
`oracle/gen_golden.py`, committed as fixtures under `tests/golden/` and checked by
`tests/test_oracle_golden.py`.  Two third-party seams (diffusers 0.29.0 Attention/GELU,
transformers Qwen2 pinned 4.40.1 vs installed 5.15.0) have no reference-held test at all:
parity there is "unpinned by the reference" (DESIGN.md §Oracle).
"""
