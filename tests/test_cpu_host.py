"""CPU-side checks (no GPU): the C ABI library loads and exports every symbol include/mmx_hip.h declares; the
state-dict manifests generated from constructor arguments equal the ones captured from the reference's modules;
the weight packing that turns Conv1d / ConvTranspose1d into windowed GEMMs is algebraically right; the product
path fails loudly without a GPU; utterance sharding + audio all-gather work across 2 gloo ranks."""
import json
import math
import os
import re
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_symbols_declared_in_header_are_exported():
    from mmx import _lib
    hdr = open(os.path.join(ROOT, "include", "mmx_hip.h")).read()
    declared = sorted(set(re.findall(r"\bint\s+(mmx_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 18
    lib = _lib.load()
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(_lib.SYMBOLS) == declared
    assert lib.mmx_abi_version() == 10


def test_gemm_params_struct_matches_header_layout():
    """ctypes mirror of MmxGemmParams: field order/names must follow the header."""
    from mmx import _lib
    hdr = open(os.path.join(ROOT, "include", "mmx_hip.h")).read()
    start = hdr.index("typedef struct MmxGemmParams {") + len("typedef struct MmxGemmParams {")
    body = hdr[start:hdr.index("} MmxGemmParams;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(re.findall(r"([A-Za-z_0-9]+)\s*$", part.strip())[0])
    assert names == [f[0] for f in _lib.GemmParams._fields_]


@pytest.mark.parametrize("name,fn", [("dac80", lambda s: s.dac_decoder_manifest(80)), ("dac128", lambda s: s.dac_decoder_manifest(128)),
                                     ("dacenc", lambda s: s.dac_encoder_manifest(80)),
                                     ("flow", lambda s: s.flow_manifest()), ("llm", lambda s: s.llm_manifest())])
def test_manifests_equal_reference_state_dicts(golden_dir, name, fn):
    from mmx import shapes
    ref = {k: tuple(v) for k, v in json.load(open(os.path.join(golden_dir, f"manifest_{name}.json"))).items()}
    assert fn(shapes) == ref


def _window_gemm(x, Wp, M, N, cin, ntaps, dil, row_off, row_hi):
    """numpy-level statement of the windowed GEMM contract (include/mmx_hip.h) for packing checks."""
    out = torch.zeros(M, N)
    for tap in range(ntaps):
        rows = torch.arange(M) + tap * dil + row_off
        ok = (rows >= 0) & (rows < row_hi)
        a = torch.zeros(M, cin)
        a[ok] = x[rows[ok]]
        out += a @ Wp[:, tap * cin:(tap + 1) * cin].t()
    return out


@pytest.mark.parametrize("k,dil", [(7, 1), (7, 9), (3, 1), (1, 1)])
def test_conv1d_packing(k, dil):
    from mmx import ops
    g = torch.Generator().manual_seed(k + dil)
    Cin, Cout, T = 16, 24, 40
    x, w = torch.randn(T, Cin, generator=g), torch.randn(Cout, Cin, k, generator=g)
    pad = (k - 1) * dil // 2
    Wp = ops.pack_conv1d(w, 0)
    got = _window_gemm(x, Wp, T, Cout, Cin, k, dil, -pad, T)
    ref = F.conv1d(x.t()[None], w, dilation=dil, padding=pad)[0].t()
    assert torch.allclose(got, ref, atol=1e-4)


@pytest.mark.parametrize("s", [2, 3, 4, 5])
def test_convtranspose1d_packing(s):
    from mmx import ops
    g = torch.Generator().manual_seed(s)
    Cin, Cout, T = 16, 8, 11
    x, w = torch.randn(T, Cin, generator=g), torch.randn(Cin, Cout, 2 * s, generator=g)
    Wp = ops.pack_convtranspose1d(w, s, 0)
    z = _window_gemm(x, Wp, T + 1, s * Cout, Cin, 2, 1, -1, T)          # rows q = 0..T, columns (k0, co)
    pad = math.ceil(s / 2)
    flat = z.reshape(-1)[pad * Cout: pad * Cout + T * s * Cout].reshape(T * s, Cout)
    ref = F.conv_transpose1d(x.t()[None], w, stride=s, padding=pad, output_padding=s % 2)[0].t()
    assert torch.allclose(flat, ref, atol=1e-4)


def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: CPU tensors are rejected before anything is launched."""
    from mmx import ops
    if torch.cuda.is_available():
        pytest.skip("this check is for the CPU-only container")
    with pytest.raises((AssertionError, RuntimeError)):
        ops.rownorm(torch.zeros(4, 8), torch.ones(8), None, 1e-5, rows=4, C_=8, out_f32=torch.zeros(4, 8))


def test_shard_utterances_is_balanced():
    from mmx.dist import shard_utterances
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(50, 501, (256,), generator=g).tolist()
    sh = shard_utterances(lens, 8)
    assert sorted(i for s in sh for i in s) == list(range(256))
    assert {len(s) for s in sh} == {32}
    tot = [sum(lens[i] for i in s) for s in sh]
    assert max(tot) / min(tot) < 1.02


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[1], "minimax-speech_amd"))
from mmx.dist import shard_utterances, gather_audio
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
lens = [300, 120, 480, 50, 77]
sh = shard_utterances(lens, 2)
mine = sh[rank]
wavs = [torch.full((lens[i] * 3,), float(i + 1)) for i in mine]
buf, n, row_of = gather_audio(wavs, mine, len(lens), 480 * 3)
for i, L in enumerate(lens):
    assert int(n[i]) == L * 3, (i, n)
    a = buf[int(row_of[i])]
    assert torch.all(a[:L * 3] == i + 1) and torch.all(a[L * 3:] == 0)
dist.destroy_process_group()
print("ok", rank)
"""


def test_two_rank_audio_all_gather_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    port = 29500 + os.getpid() % 2000
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


_BENCH_WORKER = r"""
import json, os, sys, torch
root = sys.argv[1]
sys.path.insert(0, root)
os.environ.update(RANK=sys.argv[3], LOCAL_RANK=sys.argv[3], WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2],
                  MMX_BENCH_REHEARSE="1")
import bench


class StubEngine:
    # stands for TtsEngine in the CPU rehearsal: zero waveforms of the right lengths (2 frames per token, 480 samples per frame)
    hop = 480

    def tts_batch(self, texts, embs, seed=0, exact_steps=None, **kw):
        return [torch.zeros(1, 1, 2 * n * self.hop) for n in exact_steps]


bench.make_engine = lambda dt, device, max_batch, max_ctx: StubEngine()
sys.argv = ["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--per-gpu", "4", "--no-cpu-baseline", "--no-extras"]
bench.main()
"""


def test_bench_two_rank_rehearsal_gloo(tmp_path):
    """bench.py's multi-rank contract on 2 gloo ranks (CPU tensors, stub engine): env-driven rendezvous, length-balanced
    sharding, barrier-bracketed timed region, max-over-ranks time, all-gather of the audio, ONE JSON line from rank 0
    whose value is the whole-job aggregate."""
    script = tmp_path / "b.py"
    script.write_text(_BENCH_WORKER)
    port = 31500 + os.getpid() % 2000
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE) for r in range(2)]
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1].decode()[-2000:] for o in outs]
    lines = [l for l in outs[0][0].decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].decode().splitlines() if l.startswith("{")]
    j = json.loads(lines[0])
    lens = torch.randint(50, 501, (8,), generator=torch.Generator().manual_seed(3)).tolist()
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert abs(j["config"]["audio_s_per_step"] - sum(lens) / 25.0) < 0.01          # both ranks' utterances are counted


def test_scheduler_cost_model_refit():
    """TtsEngine._refit_sched (host logic of tts_batch): the fit of (decode step, group cost = a + b * frames) from a call's own
    events; calls that captured plans are ignored; with sched_adapt the model follows two agreeing steady fits past the hysteresis."""
    from types import SimpleNamespace
    from mmx.pipeline import TtsEngine

    class Ev:
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t

    def groups(a, b):
        return [(f, True, Ev(0.0), Ev(a + b * f)) for f in (600, 1500, 2900, 4200)] + [(900, False, Ev(0.0), Ev(1e3))]   # (the last: not beside the loop)

    eng = SimpleNamespace(sched={"step_ms": 1.0, "group_ms": 40.0, "frame_ms": 0.01}, sched_fit=None, _sched_prev=None, sched_adapt=False,
                          SCHED_HYSTERESIS=TtsEngine.SCHED_HYSTERESIS)
    TtsEngine._refit_sched(eng, 500, 600.0, groups(50.0, 0.02), steady=True)
    assert abs(eng.sched_fit["step_ms"] - 1.2) < 1e-6 and abs(eng.sched_fit["group_ms"] - 50.0) < 1e-3 and abs(eng.sched_fit["frame_ms"] - 0.02) < 1e-6
    assert eng.sched == {"step_ms": 1.0, "group_ms": 40.0, "frame_ms": 0.01}                      # reported, not adopted
    eng.sched_adapt = True
    TtsEngine._refit_sched(eng, 500, 2000.0, groups(300.0, 0.05), steady=False)                 # a call that captured plans
    assert eng._sched_prev is None and eng.sched["frame_ms"] == 0.01
    TtsEngine._refit_sched(eng, 500, 600.0, groups(50.0, 0.02), steady=True)                    # first steady fit: nothing to agree with yet
    assert eng.sched["frame_ms"] == 0.01
    TtsEngine._refit_sched(eng, 500, 606.0, groups(51.0, 0.0201), steady=True)                  # second: agrees, off by more than 25 %
    assert abs(eng.sched["frame_ms"] - 0.0201) < 1e-6 and abs(eng.sched["group_ms"] - 51.0) < 1e-3
    assert eng.sched["step_ms"] == 1.0                                                          # 1.21 is within the hysteresis of 1.0


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md maps every exported entry point to the reference code it replaces; include/mmx_hip.h declares each one."""
    from mmx import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    hdr = open(os.path.join(root, "include", "mmx_hip.h")).read()
    assert [s for s in _lib.SYMBOLS if f"`{s}`" not in doc] == []
    assert [s for s in _lib.SYMBOLS if not re.search(r"\b" + s + r"\(", hdr)] == []
    assert f"mmx_abi_version() == {_lib.ABI_VERSION}" in doc


def test_flow_tile_rule_of_the_split_build():
    """FlowEngine._tile_rows (host logic, no GPU): the split build's fused tail kernel runs on the tile height with the shortest
    modelled launch (profiles/r04_tail_lab64_x.txt: 16 rows for one utterance, 32 rows up to one round of 256 workgroups, 64 rows where
    32-row tiles would need a second round) and on 64 rows beside the decode loop; the ResNet kernel never on more than 32 rows.  The
    choice never changes results (tests/test_gpu_split.py::test_split_tiles_of_every_height_agree_bit_for_bit)."""
    from mmx.flow import FlowEngine
    fl = object.__new__(FlowEngine)
    fl.dtype, fl.split, fl.wplanes, fl.polite = 2, True, False, False
    assert fl._tile_rows(2, 500) == (16, 16)              # one 10 s utterance with its CFG pair: 64 workgroups of 16 rows
    assert fl._tile_rows(8, 1000)[0] == 32                # 8 000 rows: 250 workgroups of 32 rows, one round
    assert fl._tile_rows(10, 1000) == (64, 32)            # 10 000 rows: 313 32-row tiles would take a second round
    assert fl._tile_rows(16, 896) == (64, 32)             # the step's largest group
    fl.polite = True                                      # beside the decode loop: 64 rows whatever the size
    assert fl._tile_rows(4, 700) == (64, 32)
    fl.max_tile_rows = 32                                 # bench.py --flow-bm 32
    assert fl._tile_rows(4, 700) == (32, 32)
    del fl.max_tile_rows
    fl.polite, fl.wplanes = False, True                   # weight planes: their own launch model, the same shape of rule
    assert fl._tile_rows(2, 500)[0] == 16 and fl._tile_rows(16, 896)[0] == 64
    # the model itself: full rounds of 256 workgroups, then the remainder; a fuller chip streams the shared weights slower
    assert FlowEngine._launch_us(64, 256, True) > FlowEngine._launch_us(64, 16, True) > FlowEngine._launch_us(32, 16, True)
    assert abs(FlowEngine._launch_us(32, 512, True) - 2 * FlowEngine._launch_us(32, 256, True)) < 1e-9
