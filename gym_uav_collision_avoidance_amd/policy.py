"""Inference-side glue for policies trained with the reference's SAC (pytorch_sac_temp/): the same
module/parameter names as GaussianPolicy (model.py:64-78) so `weights.chpt` files written by
SAC.save_checkpoint (sac.py:101-114, key 'policy_state_dict') load unchanged, and ONE batched forward
over all (env, agent) rows replaces the per-agent select_action round trips (sac.py:38-44).
The learners themselves are out of scope."""
import torch
import torch.nn as nn
import torch.nn.functional as F

LOG_SIG_MAX, LOG_SIG_MIN = 2, -20  # model.py:6-7


class GaussianPolicy(nn.Module):
    def __init__(self, num_inputs=10, num_actions=2, hidden=256):
        super().__init__()
        self.linear1 = nn.Linear(num_inputs, hidden)
        self.linear2 = nn.Linear(hidden, hidden)
        self.mean_linear = nn.Linear(hidden, num_actions)
        self.log_std_linear = nn.Linear(hidden, num_actions)

    def forward(self, state):
        x = F.relu(self.linear1(state))
        x = F.relu(self.linear2(x))
        return self.mean_linear(x), torch.clamp(self.log_std_linear(x), min=LOG_SIG_MIN, max=LOG_SIG_MAX)

    @torch.no_grad()
    def act(self, obs, evaluate=True, generator=None):
        """obs [..., 10] -> actions [..., 2] in [-1, 1] (sac.py:38-44: tanh(mean) when evaluating,
        tanh(mean + std*eps) otherwise).  Feed the result to step_ex(..., polar=True)."""
        mean, log_std = self.forward(obs)
        if evaluate:
            return torch.tanh(mean)
        eps = torch.randn(mean.shape, generator=generator, device=mean.device, dtype=mean.dtype)
        return torch.tanh(mean + log_std.exp() * eps)


def load_reference_checkpoint(path, device="cuda", num_inputs=10, num_actions=2):
    """Loads the policy part of a reference `weights.chpt` (tensors only: weights_only=True)."""
    ckpt = torch.load(path, map_location=device, weights_only=True)
    sd = ckpt["policy_state_dict"] if "policy_state_dict" in ckpt else ckpt
    sd = {k: v for k, v in sd.items() if k.split(".")[0] in ("linear1", "linear2", "mean_linear", "log_std_linear")}
    pol = GaussianPolicy(num_inputs, num_actions, hidden=sd["linear1.weight"].shape[0]).to(device)
    pol.load_state_dict(sd)
    return pol.eval()
