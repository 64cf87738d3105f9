"""Drop-in subset of speech/matcha/models/components/flow_matching.py: BASECFM (:12-118) is only a base class on
the CosyVoice2 path; cosyvoice.flow.flow_matching.ConditionalCFM carries the behaviour."""
import torch


class BASECFM(torch.nn.Module):
    def __init__(self, n_feats, cfm_params, n_spks=1, spk_emb_dim=128):
        super().__init__()
        self.n_feats, self.n_spks, self.spk_emb_dim = n_feats, n_spks, spk_emb_dim
        self.solver = getattr(cfm_params, "solver", "euler")
        self.sigma_min = getattr(cfm_params, "sigma_min", 1e-4)
        self.estimator = None
