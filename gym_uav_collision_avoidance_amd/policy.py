"""Inference-side glue for policies trained with the reference's learners: the same module / parameter names as their
actor networks, so that checkpoint files written by the reference load unchanged, and ONE batched forward over all
(env, agent) rows replaces the per-agent select_action round trips (sac.py:38-44, td3.py:95-97, ddpg.py:39-47):
    SAC   pytorch_sac_temp/   GaussianPolicy (model.py:64-78)  weights.chpt  'policy_state_dict'   (sac.py:101-114)
    TD3   pytorch_td3_temp/   Actor (td3.py:14-27)             weights.chpt  'actor_state_dict'    (td3.py:159-170)
    DDPG  pytorch_ddpg_temp/  ActorNetwork (model.py:6-31)     actor.chpt    'model_state_dict'    (ddpg.py:124-135)
Every actor emits a in [-1, 1]^2, which all three trainers convert the same way (test_sac_multi.py:77-80,
test_td3_multi.py:77-79, test_ddpg_multi.py:78-80): feed it to step_ex(..., polar=True).
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


# ---- TD3 (pytorch_td3_temp/td3.py) ---------------------------------------------------------------------------------
class TD3Actor(nn.Module):
    """Deterministic actor with the parameter names of td3.py:14-27: l1 (state -> 256), l2 (256 -> 256), l3 (256 -> action),
    relu / relu / tanh."""

    def __init__(self, num_inputs=10, num_actions=2, hidden=256):
        super().__init__()
        self.l1 = nn.Linear(num_inputs, hidden)
        self.l2 = nn.Linear(hidden, hidden)
        self.l3 = nn.Linear(hidden, num_actions)

    def forward(self, state):
        return torch.tanh(self.l3(F.relu(self.l2(F.relu(self.l1(state))))))

    @torch.no_grad()
    def act(self, obs, evaluate=True, noise_std=0.1, generator=None):
        """obs [..., 10] -> actions [..., 2] in [-1, 1].  evaluate=False adds the exploration noise of the reference's trainer
        (test_td3_multi.py:70-76: Gaussian on the actor output, clipped to [-1, 1])."""
        a = self.forward(obs)
        if evaluate:
            return a
        eps = torch.randn(a.shape, generator=generator, device=a.device, dtype=a.dtype)
        return (a + noise_std * eps).clamp_(-1.0, 1.0)


class TD3TwinQ(nn.Module):
    """Parameter container with the names of the TD3 critic (td3.py:29-54: l1..l3 = Q1, l4..l6 = Q2 on [state, action])."""

    def __init__(self, num_inputs=10, num_actions=2, hidden=256):
        super().__init__()
        self.l1 = nn.Linear(num_inputs + num_actions, hidden)
        self.l2 = nn.Linear(hidden, hidden)
        self.l3 = nn.Linear(hidden, 1)
        self.l4 = nn.Linear(num_inputs + num_actions, hidden)
        self.l5 = nn.Linear(hidden, hidden)
        self.l6 = nn.Linear(hidden, 1)

    def forward(self, state, action):
        sa = torch.cat([state, action], dim=-1)
        return (self.l3(F.relu(self.l2(F.relu(self.l1(sa))))), self.l6(F.relu(self.l5(F.relu(self.l4(sa))))))


TD3_CHECKPOINT_KEYS = ("actor_state_dict", "actor_target_state_dict", "critic_state_dict", "critic_target_state_dict",
                       "actor_optimizer_state_dict", "critic_optimizer_state_dict")   # td3.py:163-169


def load_td3_checkpoint(path, device="cuda", num_inputs=10, num_actions=2):
    """Loads the actor of a reference TD3 `weights.chpt` (tensors only: weights_only=True)."""
    ckpt = torch.load(path, map_location=device, weights_only=True)
    sd = ckpt["actor_state_dict"] if "actor_state_dict" in ckpt else ckpt
    sd = {k: v for k, v in sd.items() if k.split(".")[0] in ("l1", "l2", "l3")}
    pol = TD3Actor(num_inputs, num_actions, hidden=sd["l1.weight"].shape[0]).to(device)
    pol.load_state_dict(sd)
    return pol.eval()


def _cpu_sd(sd):
    return {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in sd.items()}


def _cpu_opt(sd):
    state = {k: {kk: (vv.detach().cpu() if torch.is_tensor(vv) else vv) for kk, vv in st.items()} for k, st in sd["state"].items()}
    return {"state": state, "param_groups": sd["param_groups"]}


def save_td3_checkpoint(path, actor, critic=None, actor_optimizer=None, critic_optimizer=None, lr=3e-4):
    """Writes `path` with the six keys of TD3.save_checkpoint (td3.py:159-170), targets = copies (copy.deepcopy at td3.py:80,84),
    Adam(lr = 3e-4) optimisers (td3.py:81,85) unless given: TD3.load_checkpoint's load_state_dict calls accept every entry."""
    import os
    n_in, n_act, hidden = actor.l1.in_features, actor.l3.out_features, actor.l1.out_features
    critic = critic if critic is not None else TD3TwinQ(n_in, n_act, hidden)
    actor_optimizer = actor_optimizer if actor_optimizer is not None else torch.optim.Adam(actor.parameters(), lr=lr)
    critic_optimizer = critic_optimizer if critic_optimizer is not None else torch.optim.Adam(critic.parameters(), lr=lr)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save({"actor_state_dict": _cpu_sd(actor.state_dict()), "actor_target_state_dict": _cpu_sd(actor.state_dict()),
                "critic_state_dict": _cpu_sd(critic.state_dict()), "critic_target_state_dict": _cpu_sd(critic.state_dict()),
                "actor_optimizer_state_dict": _cpu_opt(actor_optimizer.state_dict()),
                "critic_optimizer_state_dict": _cpu_opt(critic_optimizer.state_dict())}, path)
    return path


# ---- DDPG (pytorch_ddpg_temp/) ---------------------------------------------------------------------------------------
class DDPGActor(nn.Module):
    """Deterministic actor with the module names of pytorch_ddpg_temp/model.py:6-31: input (state -> 400), fc1 (400 -> 300),
    fc2 (300 -> action), LeakyReLU / LeakyReLU / tanh.  The reference's module also owns an unused BatchNorm1d `bn1` (400
    features, affine=False, commented out of its forward): it is kept so that the three buffers it contributes to the state
    dict (running_mean, running_var, num_batches_tracked) load and save like the reference's."""

    def __init__(self, num_inputs=10, num_actions=2, hidden1=400, hidden2=300):
        super().__init__()
        self.input = nn.Linear(num_inputs, hidden1)
        self.fc1 = nn.Linear(hidden1, hidden2)
        self.fc2 = nn.Linear(hidden2, num_actions)
        self.bn1 = nn.BatchNorm1d(num_features=hidden1, eps=0.001, momentum=0.01, affine=False)

    def forward(self, state):
        x = F.leaky_relu(self.input(state))
        x = F.leaky_relu(self.fc1(x))
        return torch.tanh(self.fc2(x))

    @torch.no_grad()
    def act(self, obs, evaluate=True, noise=None):
        """obs [..., 10] -> actions [..., 2] in [-1, 1].  evaluate=False: `noise` (a tensor broadcastable to the output, e.g.
        the caller's Ornstein-Uhlenbeck state: ddpg.py:39-47 adds OUActionNoise and clips to [-1, 1]) is added and clipped."""
        a = self.forward(obs)
        if evaluate or noise is None:
            return a
        return (a + noise).clamp_(-1.0, 1.0)


class DDPGCritic(nn.Module):
    """Parameter container with the names of the DDPG critic (model.py:36-57: input, fc1, fc2 on [state, action])."""

    def __init__(self, num_inputs=10, num_actions=2, hidden1=400, hidden2=300):
        super().__init__()
        self.input = nn.Linear(num_inputs + num_actions, hidden1)
        self.fc1 = nn.Linear(hidden1, hidden2)
        self.fc2 = nn.Linear(hidden2, 1)

    def forward(self, state, action):
        x = F.leaky_relu(self.input(torch.cat([state, action], dim=-1)))
        return self.fc2(F.leaky_relu(self.fc1(x)))


DDPG_CHECKPOINT_KEYS = ("model_state_dict", "target_model_state_dict", "optimizer_state_dict")   # ddpg.py:125-135, both files


def load_ddpg_checkpoint(path, device="cuda", num_inputs=10, num_actions=2):
    """Loads the actor of a reference DDPG checkpoint: `path` is the directory DDPG.save_checkpoint wrote (ddpg.py:124-135) or
    its actor.chpt itself (tensors only: weights_only=True)."""
    import os
    f = os.path.join(path, "actor.chpt") if os.path.isdir(path) else path
    ckpt = torch.load(f, map_location=device, weights_only=True)
    sd = ckpt["model_state_dict"] if "model_state_dict" in ckpt else ckpt
    pol = DDPGActor(num_inputs, num_actions, hidden1=sd["input.weight"].shape[0], hidden2=sd["fc1.weight"].shape[0]).to(device)
    pol.load_state_dict(sd)
    return pol.eval()


def save_ddpg_checkpoint(directory, actor, critic=None, actor_optimizer=None, critic_optimizer=None, actor_lr=1e-4, critic_lr=1e-3):
    """Writes actor.chpt and critic.chpt into `directory` with the three keys each that DDPG.save_checkpoint uses
    (ddpg.py:124-135); targets = copies (_hard_update, ddpg.py:27-28), Adam(amsgrad=True) optimisers (ddpg.py:21,25)."""
    import os
    n_in, n_act = actor.input.in_features, actor.fc2.out_features
    critic = critic if critic is not None else DDPGCritic(n_in, n_act, actor.input.out_features, actor.fc1.out_features)
    actor_optimizer = actor_optimizer if actor_optimizer is not None else torch.optim.Adam(actor.parameters(), lr=actor_lr, amsgrad=True)
    critic_optimizer = critic_optimizer if critic_optimizer is not None else torch.optim.Adam(critic.parameters(), lr=critic_lr, amsgrad=True)
    os.makedirs(directory, exist_ok=True)
    torch.save({"model_state_dict": _cpu_sd(actor.state_dict()), "target_model_state_dict": _cpu_sd(actor.state_dict()),
                "optimizer_state_dict": _cpu_opt(actor_optimizer.state_dict())}, os.path.join(directory, "actor.chpt"))
    torch.save({"model_state_dict": _cpu_sd(critic.state_dict()), "target_model_state_dict": _cpu_sd(critic.state_dict()),
                "optimizer_state_dict": _cpu_opt(critic_optimizer.state_dict())}, os.path.join(directory, "critic.chpt"))
    return directory


def load_actor(path, kind=None, device="cuda"):
    """Loads whichever of the three reference actors `path` holds (kind: "sac" / "td3" / "ddpg", or None to tell from the keys)."""
    import os
    if kind is None:
        f = os.path.join(path, "actor.chpt") if os.path.isdir(path) else path
        keys = set(torch.load(f, map_location="cpu", weights_only=True))
        kind = "sac" if "policy_state_dict" in keys else "td3" if "actor_state_dict" in keys else "ddpg" if "model_state_dict" in keys else None
    if kind == "sac":
        return load_reference_checkpoint(path, device=device)
    if kind == "td3":
        return load_td3_checkpoint(path, device=device)
    if kind == "ddpg":
        return load_ddpg_checkpoint(path, device=device)
    raise ValueError(f"uavx: {path} is not a SAC / TD3 / DDPG checkpoint of the reference's layouts")
