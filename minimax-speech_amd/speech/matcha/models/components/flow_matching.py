"""Drop-in subset of speech/matcha/models/components/flow_matching.py: BASECFM (:12-118), the plain (no CFG) Euler
solver base class.  On the CosyVoice2 path cosyvoice.flow.flow_matching.ConditionalCFM carries the behaviour; this class
keeps the base API working on the HIP kernels: `estimator` is any module with the estimator seam signature."""
import torch

from ... import _paths  # noqa: F401


class BASECFM(torch.nn.Module):
    def __init__(self, n_feats, cfm_params, n_spks=1, spk_emb_dim=128):
        super().__init__()
        self.n_feats, self.n_spks, self.spk_emb_dim = n_feats, n_spks, spk_emb_dim
        g = (lambda k, d: cfm_params.get(k, d)) if isinstance(cfm_params, dict) else (lambda k, d: getattr(cfm_params, k, d))
        self.solver = g("solver", "euler")
        self.sigma_min = g("sigma_min", 1e-4)
        self.estimator = None

    @torch.inference_mode()
    def forward(self, mu, mask, n_timesteps, temperature=1.0, spks=None, cond=None):
        """flow_matching.py:33-52: z ~ N(0, temperature^2) -> solve_euler over linspace(0, 1, n_timesteps + 1)."""
        z = torch.randn_like(mu) * temperature
        t_span = torch.linspace(0, 1, n_timesteps + 1, device=mu.device)
        return self.solve_euler(z, t_span=t_span, mu=mu, mask=mask, spks=spks, cond=cond)

    @torch.inference_mode()
    def solve_euler(self, x, t_span, mu, mask, spks, cond):
        """flow_matching.py:54-84: x += dt * estimator(x, mask, mu, t, spks, cond), fixed steps over t_span."""
        from mmx import ops
        x = x.to(torch.float32).contiguous().clone()
        t, dt = t_span[0], t_span[1] - t_span[0]
        for step in range(1, len(t_span)):
            d = self.estimator(x, mask, mu, t.reshape(1).expand(x.shape[0]).contiguous(), spks, cond).contiguous()
            ops.cfg_euler(x, d, d, 0.0, float(dt), x.numel())
            t = t + dt
            if step < len(t_span) - 1:
                dt = t_span[step + 1] - t
        return x
