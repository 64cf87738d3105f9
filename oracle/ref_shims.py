"""TEST INFRASTRUCTURE ONLY (oracle/): import shims for running the *reference's own*
Python classes in this build container, to (a) validate the CPU restatement in
oracle/*.py and (b) generate the golden vectors committed under tests/golden/.

Never imported by the product path, by bench.py's timed region or on the GPU box
(/root/reference does not exist there).

The reference hot path imports third-party modules that are absent offline
(SURVEY.md §8c).  Pure plumbing ones (torchaudio, lightning, hydra, onnxruntime ...)
are replaced by empty stand-ins.  Two carry arithmetic and are restated here from
their published, pinned versions (parity at that seam is therefore "unpinned by the
reference", as SURVEY.md §8c records):

  * diffusers==0.29.0 (requirements.txt:5)
      models.attention_processor.Attention  (AttnProcessor2_0 path, self-attention)
      models.attention.GELU                 (Linear + exact erf GELU)
      models.lora.LoRACompatibleLinear      (== nn.Linear without a LoRA layer)
      models.activations.get_activation     ("silu", "mish", "gelu", "swish")
    call sites: speech/matcha/models/components/transformer.py:5-14,110,126,196-204
                speech/matcha/models/components/decoder.py:8,92
  * omegaconf.DictConfig -> attribute dict (config container only)

Always run with sys.dont_write_bytecode = True so nothing is written into
/root/reference.
"""
import math
import sys
import types
import importlib.machinery

sys.dont_write_bytecode = True

REF = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []  # behave like a package so "import a.b" works
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Anything:
    """Placeholder class/callable for symbols that are imported but never executed."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, k):
        return _Anything()


def _lenient(k):
    if k.startswith("__"):
        raise AttributeError(k)
    return _Anything


def _stub_tree(root, names=()):
    m = _mod(root)
    m.__getattr__ = _lenient  # type: ignore
    for n in names:
        sub = _mod(root + "." + n)
        sub.__getattr__ = _lenient  # type: ignore
        setattr(m, n.split(".")[0], sys.modules[root + "." + n.split(".")[0]])
    return m


def install_common():
    import torch  # noqa: F401
    import transformers  # noqa: F401  (must be imported before torchaudio is stubbed)
    from transformers import Qwen2ForCausalLM  # noqa: F401  (resolve the lazy import now)

    class DictConfig(dict):
        def __init__(self, content=None, **kw):
            super().__init__(content or {}, **kw)

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

    _mod("omegaconf", DictConfig=DictConfig)
    for root, subs in [
        ("torchaudio", ["transforms", "compliance", "compliance.kaldi", "functional"]),
        ("lightning", ["pytorch", "pytorch.utilities"]),
        ("hydra", ["utils"]),
        ("onnxruntime", []),
        ("whisper", []),
        ("tiktoken", []),
        ("modelscope", []),
        ("deepspeed", []),
        ("hyperpyyaml", []),
        ("librosa", ["filters"]),
        ("soundfile", []),
        ("julius", []),
        ("flatten_dict", []),
        ("inflect", []),
        ("pyworld", []),
        ("conformer", []),
        ("tensorrt", []),
        ("tqdm", []),
    ]:
        if root in sys.modules and root == "tqdm":
            continue
        try:
            __import__(root)
            continue
        except Exception:
            pass
        _stub_tree(root, subs)
    sys.modules["lightning.pytorch.utilities"].rank_zero_only = lambda f: f


def install_diffusers():
    """diffusers 0.29.0 restatement (see module docstring)."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    class LoRACompatibleLinear(nn.Linear):
        pass

    class GELU(nn.Module):
        def __init__(self, dim_in, dim_out, approximate="none", bias=True):
            super().__init__()
            self.proj = nn.Linear(dim_in, dim_out, bias=bias)
            self.approximate = approximate

        def forward(self, x):
            return F.gelu(self.proj(x), approximate=self.approximate)

    class Attention(nn.Module):
        def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0,
                     bias=False, upcast_attention=False, **kw):
            super().__init__()
            assert cross_attention_dim is None
            inner = heads * dim_head
            self.heads, self.dim_head = heads, dim_head
            self.scale = dim_head ** -0.5
            self.to_q = nn.Linear(query_dim, inner, bias=bias)
            self.to_k = nn.Linear(query_dim, inner, bias=bias)
            self.to_v = nn.Linear(query_dim, inner, bias=bias)
            self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(dropout)])

        def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
            assert encoder_hidden_states is None
            b, t, _ = hidden_states.shape
            h, d = self.heads, self.dim_head
            q = self.to_q(hidden_states).view(b, t, h, d).transpose(1, 2)
            k = self.to_k(hidden_states).view(b, t, h, d).transpose(1, 2)
            v = self.to_v(hidden_states).view(b, t, h, d).transpose(1, 2)
            if attention_mask is not None:
                # prepare_attention_mask: [B,Tq,Tk] -> repeat_interleave(heads) -> [B,H,Tq,Tk]
                attention_mask = attention_mask.repeat_interleave(h, dim=0).view(b, h, -1, attention_mask.shape[-1])
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
            o = o.transpose(1, 2).reshape(b, t, h * d).to(q.dtype)
            o = self.to_out[0](o)
            return self.to_out[1](o)

    def get_activation(name):
        return {"silu": nn.SiLU, "swish": nn.SiLU, "mish": nn.Mish, "gelu": nn.GELU, "relu": nn.ReLU}[name.lower()]()

    _stub_tree("diffusers", ["models", "models.attention", "models.attention_processor", "models.lora",
                             "models.activations", "utils", "utils.torch_utils"])
    a = sys.modules["diffusers.models.attention"]
    a.GELU = GELU
    for n in ("GEGLU", "AdaLayerNorm", "AdaLayerNormZero", "ApproximateGELU"):
        setattr(a, n, _Anything)
    sys.modules["diffusers.models.attention_processor"].Attention = Attention
    sys.modules["diffusers.models.lora"].LoRACompatibleLinear = LoRACompatibleLinear
    sys.modules["diffusers.models.activations"].get_activation = get_activation
    sys.modules["diffusers.utils.torch_utils"].maybe_allow_in_graph = lambda c: c


def import_dac():
    """Returns the reference's dac-vae `model` module (Decoder, DACVAE ...)."""
    import torch.nn as nn
    install_common()
    _stub_tree("audiotools", ["ml"])
    sys.modules["audiotools"].AudioSignal = _Anything
    sys.modules["audiotools"].STFTParams = _Anything
    sys.modules["audiotools.ml"].BaseModel = nn.Module
    sys.modules["audiotools"].ml = sys.modules["audiotools.ml"]

    class CodecMixin:
        pass

    _mod("base", CodecMixin=CodecMixin)
    p = REF + "/dac-vae"
    if p not in sys.path:
        sys.path.insert(0, p)
    import importlib
    for n in ("layers", "model"):
        sys.modules.pop(n, None)
    return importlib.import_module("model")


def import_cosyvoice():
    """Makes `cosyvoice.*` / `matcha.*` of the reference importable."""
    install_common()
    install_diffusers()
    p = REF + "/speech"
    if p not in sys.path:
        sys.path.insert(0, p)
    # matcha/utils/__init__ pulls hydra/lightning/rich at import: register a bare package
    import os
    m = _mod("matcha.utils") if False else None
    import importlib
    matcha = importlib.import_module("matcha")
    mu = types.ModuleType("matcha.utils")
    mu.__path__ = [os.path.join(p, "matcha", "utils")]
    mu.__spec__ = importlib.machinery.ModuleSpec("matcha.utils", None, is_package=True)
    sys.modules["matcha.utils"] = mu
    pl = types.ModuleType("matcha.utils.pylogger")
    import logging
    pl.get_pylogger = lambda name=__name__: logging.getLogger(name)
    sys.modules["matcha.utils.pylogger"] = pl
    legacy_kv_cache_indexing()
    return matcha


def legacy_kv_cache_indexing():
    """The reference pins transformers 4.40 (speech/requirements.txt), whose KV cache indexes as cache[layer] ->
    (key, value); llm.py:820 relies on it (`cache[0][0].size(2)`).  The installed transformers dropped
    DynamicCache.__getitem__: restore that accessor (environment compatibility only, no reference logic)."""
    from transformers import DynamicCache
    if not hasattr(DynamicCache, "__getitem__"):
        DynamicCache.__getitem__ = lambda self, i: (self.layers[i].keys, self.layers[i].values)
