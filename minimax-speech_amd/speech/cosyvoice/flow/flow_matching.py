"""Drop-in for speech/cosyvoice/flow/flow_matching.py: ConditionalCFM (:21-155) / CausalConditionalCFM (:317-348)."""
import torch

from .. import _paths  # noqa: F401
from ..utils.common import set_all_random_seed
from mmx.shell import EngineHost


class ConditionalCFM(EngineHost):
    def __init__(self, in_channels, cfm_params, n_spks=1, spk_emb_dim=64, estimator: torch.nn.Module = None):
        super().__init__()
        g = (lambda k, d=None: cfm_params[k] if k in cfm_params else d) if isinstance(cfm_params, dict) else \
            (lambda k, d=None: getattr(cfm_params, k, d))
        self.n_feats, self.n_spks, self.spk_emb_dim = in_channels, n_spks, spk_emb_dim
        self.solver, self.sigma_min = g("solver", "euler"), g("sigma_min", 1e-6)
        self.t_scheduler = g("t_scheduler", "cosine")
        self.training_cfg_rate, self.inference_cfg_rate = g("training_cfg_rate", 0.2), g("inference_cfg_rate", 0.7)
        assert self.t_scheduler == "cosine", "config.yaml:96: cosine schedule"
        self.estimator = estimator

    def forward_estimator(self, x, mask, mu, t, spks, cond, streaming=False):
        """flow_matching.py:128-131 (the nn.Module branch; the TensorRT branch has no counterpart here)."""
        return self.estimator(x, mask, mu, t, spks, cond, streaming=streaming)

    @torch.inference_mode()
    def solve_euler(self, x, t_span, mu, mask, spks, cond, streaming=False):
        """flow_matching.py:74-126: fixed-step Euler over t_span with classifier-free guidance — per step the CFG pair
        [cond; uncond] goes through forward_estimator, then x += dt * ((1 + cfg) * d_cond - cfg * d_uncond) (one kernel).
        x, mu, cond [1, 80, T]; mask [1, 1, T]; spks [1, 80].  `forward` uses the recorded whole-solve graph instead."""
        from mmx import ops
        dev = x.device
        T = x.size(2)
        x = x.to(torch.float32).contiguous().clone()
        t, dt = t_span[0:1].to(dev, torch.float32), (t_span[1] - t_span[0]).to(dev, torch.float32)
        x_in, mask_in = torch.zeros(2, 80, T, device=dev), torch.zeros(2, 1, T, device=dev)
        mu_in, t_in = torch.zeros(2, 80, T, device=dev), torch.zeros(2, device=dev)
        spks_in, cond_in = torch.zeros(2, 80, device=dev), torch.zeros(2, 80, T, device=dev)
        for step in range(1, len(t_span)):
            x_in[:] = x
            mask_in[:] = mask
            mu_in[0] = mu
            t_in[:] = t
            spks_in[0] = spks
            cond_in[0] = cond
            d = self.forward_estimator(x_in, mask_in, mu_in, t_in, spks_in, cond_in, streaming).contiguous()
            ops.cfg_euler(x, d[0], d[1], self.inference_cfg_rate, float(dt), 80 * T)
            t = t + dt
            if step < len(t_span) - 1:
                dt = (t_span[step + 1].to(dev) - t).reshape(())
        return x.float()


class CausalConditionalCFM(ConditionalCFM):
    def __init__(self, in_channels, cfm_params, n_spks=1, spk_emb_dim=64, estimator: torch.nn.Module = None):
        super().__init__(in_channels, cfm_params, n_spks, spk_emb_dim, estimator)
        set_all_random_seed(0)                             # flow_matching.py:320: reseeds ALL global RNGs
        self.rand_noise = torch.randn([1, 80, 50 * 300])   # :321, part of the model's observable behaviour

    @torch.inference_mode()
    def forward(self, mu, mask, n_timesteps, temperature=1.0, spks=None, cond=None, streaming=False):
        """mu, cond [1,80,T]; mask [1,1,T]; spks [1,80] -> ([1,80,T] fp32, None)   (flow_matching.py:323-348)."""
        assert temperature == 1.0 and mu.shape[0] == 1
        eng = self.estimator._eng()
        eng.n_timesteps, eng.cfg = n_timesteps, self.inference_cfg_rate
        eng.set_noise(self.rand_noise)
        T = mu.shape[2]
        tm = lambda a: eng._to_time_major(a.to(eng.dev, torch.float32).contiguous(), 1, T)[0]
        x = eng.cfm(tm(mu), spks.to(eng.dev, torch.float32).reshape(-1), tm(cond), streaming)
        out = torch.empty(1, 80, T, dtype=torch.float32, device=eng.dev)
        from mmx import ops
        from mmx._lib import F32
        ops.copy2d(x, F32, 0, 80, 1, out, F32, 0, 1, T, rows=T, cols=80)
        return out, None
