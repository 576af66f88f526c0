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


class TwinQ(nn.Module):
    """Parameter container with the module names of the reference's twin critic (pytorch_sac_temp/model.py:34-48:
    linear1..3 = Q1, linear4..6 = Q2 on [state, action]), so that its state_dict is what SAC.load_checkpoint expects
    under 'critic_state_dict' / 'critic_target_state_dict'.  The learner itself is out of scope."""

    def __init__(self, num_inputs=10, num_actions=2, hidden=256):
        super().__init__()
        self.linear1 = nn.Linear(num_inputs + num_actions, hidden)
        self.linear2 = nn.Linear(hidden, hidden)
        self.linear3 = nn.Linear(hidden, 1)
        self.linear4 = nn.Linear(num_inputs + num_actions, hidden)
        self.linear5 = nn.Linear(hidden, hidden)
        self.linear6 = nn.Linear(hidden, 1)
        for m in self.modules():                      # model.py:10-14
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight, gain=1)
                nn.init.constant_(m.bias, 0)

    def forward(self, state, action):
        xu = torch.cat([state, action], dim=-1)
        q1 = self.linear3(F.relu(self.linear2(F.relu(self.linear1(xu)))))
        q2 = self.linear6(F.relu(self.linear5(F.relu(self.linear4(xu)))))
        return q1, q2


CHECKPOINT_KEYS = ("policy_state_dict", "critic_state_dict", "critic_target_state_dict", "critic_optimizer_state_dict",
                   "policy_optimizer_state_dict")   # sac.py:108-112


def save_reference_checkpoint(path, policy, critic=None, critic_target=None, critic_optimizer=None, policy_optimizer=None,
                              lr=3e-4):
    """Writes `path` in the layout SAC.save_checkpoint uses (sac.py:101-114: the five keys above), so that the
    reference's SAC.load_checkpoint (sac.py:117-139) accepts it -- e.g. a policy driven / tuned against the batched env
    goes back into the reference's training scripts.  Parts the caller does not pass are fresh ones of the reference's
    shapes: a xavier-initialised twin critic (its target a copy, like hard_update at sac.py:26) and Adam(lr) optimisers
    with empty state.  Tensors are stored on the CPU."""
    import os
    n_in, n_act = policy.linear1.in_features, policy.mean_linear.out_features
    hidden = policy.linear1.out_features
    if critic is None:
        critic = TwinQ(n_in, n_act, hidden)
    if critic_target is None:
        critic_target = TwinQ(n_in, n_act, hidden)
        critic_target.load_state_dict(critic.state_dict())
    if critic_optimizer is None:
        critic_optimizer = torch.optim.Adam(critic.parameters(), lr=lr)
    if policy_optimizer is None:
        policy_optimizer = torch.optim.Adam(policy.parameters(), lr=lr)
    cpu = lambda sd: {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in sd.items()}

    def opt_cpu(sd):
        state = {k: {kk: (vv.detach().cpu() if torch.is_tensor(vv) else vv) for kk, vv in st.items()} for k, st in sd["state"].items()}
        return {"state": state, "param_groups": sd["param_groups"]}

    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    torch.save({"policy_state_dict": cpu(policy.state_dict()),
                "critic_state_dict": cpu(critic.state_dict()),
                "critic_target_state_dict": cpu(critic_target.state_dict()),
                "critic_optimizer_state_dict": opt_cpu(critic_optimizer.state_dict()),
                "policy_optimizer_state_dict": opt_cpu(policy_optimizer.state_dict())}, path)
    return path
