"""ctypes binding of libmmx_hip.so (include/mmx_hip.h).  Thin: tensors in, raw pointers out."""
import ctypes as C
import os

import torch

F32, BF16 = 0, 1
# the split builds (include/mmx_hip.h MMX_X2 / MMX_X3): weights stored and streamed as bf16, activations stored as fp32
# and split into 2 / 3 bf16 terms inside every MFMA product.  X2 is what the flow and the DAC run (16 significant bits
# per activation), X3 the LM (24 bits: its sampler turns a 1e-3 log-prob error into another token id).
X2, X3 = 2, 3
# weight planes (include/mmx_hip.h MMX_X2W / MMX_X3W; GEMM entry points only): the weights of an fp32 checkpoint as 2 / 3 bf16
# planes hi + [mid +] lo = w.  Engines built with wplanes=True pack their weights this way (ops.pack_* with these codes return
# ops.Planed tensors) and the GEMM wrappers switch the dtype code when they are handed a Planed weight; every other entry point
# keeps seeing X2 / X3.
X2W, X3W = 0x12, 0x13
# fp16 planes of the LM decode step (include/mmx_hip.h MMX_H2 / MMX_H2W; mmx_skinny2, mmx_decode_prep, mmx_decode_attn only)
H2, H2W = 4, 0x14
ACT = {"none": 0, "lrelu": 1, "gelu": 2, "silu": 3, "mish": 4, "tanh": 5}
TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16, X2: torch.float32, X3: torch.float32, X2W: torch.float32, X3W: torch.float32}       # activation storage
WEIGHT_DT = {F32: torch.float32, BF16: torch.bfloat16, X2: torch.bfloat16, X3: torch.bfloat16, X2W: torch.bfloat16, X3W: torch.bfloat16}   # weight storage
ESIZE = {F32: 4, BF16: 2, X2: 4, X3: 4}
WPLANES = {X2W: 2, X3W: 3}                               # bf16 planes per weight
DTYPE_NAMES = {"f32": F32, "bf16": BF16, "x": X2, "x2": X2, "x3": X3}


def is_split(dtype):
    return dtype in (X2, X3, X2W, X3W)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMX_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libmmx_hip.so")   # MMX_LIB: A/B of two builds

SYMBOLS = [
    "mmx_abi_version", "mmx_gemm_win", "mmx_gemm_win_tile", "mmx_rownorm", "mmx_groupnorm", "mmx_act_rows", "mmx_gather_rows", "mmx_copy2d", "mmx_est_pack",
    "mmx_sinusoidal_emb", "mmx_cfg_euler", "mmx_attn_dense", "mmx_attn_flash_bf16", "mmx_attn_flash_fp8", "mmx_attn_relpos_bf16", "mmx_attn_relpos_x", "mmx_attn_flash_x", "mmx_attn_flash_xs", "mmx_conv_cout1_tanh", "mmx_conv_cin1", "mmx_vae_sample", "mmx_resample_linear", "mmx_prefetch4", "mmx_mask_rows",
    "mmx_est_tail", "mmx_est_resnet", "mmx_dac_ru", "mmx_pack_skinny", "mmx_skinny_gemm", "mmx_skinny2", "mmx_decode_prep", "mmx_rope_kv_store", "mmx_paged_attn", "mmx_decode_attn", "mmx_swiglu", "mmx_sample_step",
]


class GemmParams(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p),
        ("rowmask", C.c_void_p), ("alpha", C.c_void_p), ("out_f32", C.c_void_p), ("out_act", C.c_void_p),
        ("lda", C.c_int64), ("ldw", C.c_int64), ("ldr", C.c_int64), ("ldo_f", C.c_int64), ("ldo_a", C.c_int64),
        ("a_bstride", C.c_int64), ("w_bstride", C.c_int64), ("r_bstride", C.c_int64), ("rm_bstride", C.c_int64),
        ("of_bstride", C.c_int64), ("oa_bstride", C.c_int64),
        ("row_off", C.c_int64), ("row_lo", C.c_int64), ("row_hi", C.c_int64),
        ("out_off", C.c_int64), ("out_len", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("batch", C.c_int32),
        ("ntaps", C.c_int32), ("cin", C.c_int32), ("dil", C.c_int32),
        ("bias_mod", C.c_int32), ("alpha_mod", C.c_int32), ("bias_per_row", C.c_int32),
        ("act", C.c_int32), ("act2", C.c_int32), ("slope", C.c_float), ("row_stride", C.c_int32),
    ]


class EstNext(C.Structure):
    _fields_ = [("wqkv", C.c_void_p), ("n1g", C.c_void_p), ("n1b", C.c_void_p), ("q_out", C.c_void_p), ("vt_out", C.c_void_p),
                ("q_bs", C.c_int64), ("vt_bs", C.c_int64), ("ldq", C.c_int32), ("ldvt", C.c_int32)]


class EstTailParams(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("ao", "x", "wo", "w1", "w2", "bo", "b1", "b2", "n3g", "n3b", "rowmask", "act_out")] + \
               [(k, C.c_int64) for k in ("ao_bs", "x_bs", "rm_bs", "act_bs")] + \
               [(k, C.c_int32) for k in ("ldao", "act_ld", "B", "T", "t_begin")] + [("eps", C.c_float), ("next", EstNext)]


class EstResnetParams(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("a_in", "x", "w1", "w2", "wr", "b1", "g1", "be1", "b2", "g2", "be2", "br", "tv", "rowmask")] + \
               [(k, C.c_int64) for k in ("a_bs", "x_bs", "tv_bs", "rm_bs")] + \
               [(k, C.c_int32) for k in ("lda", "cin", "B", "T", "t_begin")] + [("eps", C.c_float), ("next", EstNext)]


class DacRuParams(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("x", "x_out", "act_out", "w7", "w1", "b7", "b1", "a0", "a2", "alpha_next", "lens")] + \
               [("x_bs", C.c_int64)] + [(k, C.c_int32) for k in ("B", "T", "C", "dil")] + [("slope", C.c_float)]


def fill_struct(st, **kw):
    """Sets fields of a ctypes structure; tensors become device addresses."""
    for k, v in kw.items():
        if v is not None and hasattr(v, "data_ptr"):
            assert v.is_cuda
            v = v.data_ptr()
        setattr(st, k, v)
    return st


ABI_VERSION = 10                                         # include/mmx_hip.h: mmx_abi_version()
_lib = None


class MmxError(RuntimeError):
    pass


def load():
    """Loads the shared library (fails loudly: there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MmxError(f"{LIB_PATH} not found: build it with `make -C minimax-speech_amd/csrc` "
                           "(or __graft_entry__.build()); the hot path has no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for s in SYMBOLS:
            getattr(lib, s).restype = C.c_int
        if lib.mmx_abi_version() != ABI_VERSION:         # a stale build: argument lists differ, refuse it
            raise MmxError(f"{LIB_PATH} has ABI version {lib.mmx_abi_version()}, this package binds version {ABI_VERSION}: "
                           "rebuild it (`make -C minimax-speech_amd/csrc`)")
        _lib = lib
    return _lib


def _p(t):
    if t is None:
        return None
    if isinstance(t, int):
        return C.c_void_p(t)
    assert t.is_cuda, "libmmx_hip works on device memory only (no CPU fallback)"
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise MmxError(f"{what} failed with code {rc}" + (" (argument/shape error)" if rc == -1 else ""))


def dt_of(t):
    return BF16 if t.dtype == torch.bfloat16 else F32


def i64(x):
    return C.c_int64(int(x))


def gemm_params(**kw):
    p = GemmParams()
    p.bias_mod = 1
    p.alpha_mod = 1
    p.ntaps = 1
    p.dil = 1
    p.batch = 1
    p.row_lo = 0
    p.out_len = 1 << 62
    p.slope = 0.1
    p.row_stride = 1
    for k, v in kw.items():
        if k in ("A", "W", "bias", "residual", "rowmask", "alpha", "out_f32", "out_act"):
            v = None if v is None else (v if isinstance(v, int) else v.data_ptr())
        setattr(p, k, v)
    return p


TILES = {"auto": 0, "128x128": 1, "128x64": 2, "64x64": 3, "32x64": 4}


def gemm_win(p, dtype, tile=0):
    """tile != 0 forces the block tile (mmx_gemm_win_tile; tuning tools only)."""
    if tile:
        check(load().mmx_gemm_win_tile(C.byref(p), C.c_int(dtype), C.c_int(tile), stream()), "mmx_gemm_win_tile")
    else:
        check(load().mmx_gemm_win(C.byref(p), C.c_int(dtype), stream()), "mmx_gemm_win")
