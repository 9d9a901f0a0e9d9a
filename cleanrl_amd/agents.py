"""The reference's three ``Agent`` networks with identical structure, parameter names, initialisation
order (hence identical weights for a given torch seed) and method surface:

* ``AtariAgent``      -- NatureCNN, ppo_atari_multigpu.py:133-159 (== ppo_atari.py, ppo_atari_envpool.py)
* ``MlpAgent``        -- 64-64 tanh actor/critic, ppo.py:100-126
* ``ContinuousAgent`` -- 64-64 tanh + state-independent logstd, ppo_continuous_action.py:112-141

``get_value`` / ``get_action_and_value`` keep the reference signatures.  On a CUDA (HIP) device the
distribution math (sample / log_prob / entropy) runs in the libmi355ppo kernels; on an explicit CPU
device (``--no-cuda``, config A plumbing, gloo tests) it is ``torch.distributions`` exactly as in the
reference.  The learner's update does not go through these methods: it feeds the network outputs to the
fused loss kernel (``heads``/``dist_params`` below are the seam).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.distributions.categorical import Categorical
from torch.distributions.normal import Normal

from . import ops


def layer_init(layer, std=np.sqrt(2), bias_const=0.0):
    torch.nn.init.orthogonal_(layer.weight, std)
    torch.nn.init.constant_(layer.bias, bias_const)
    return layer


class _SampleCounter:
    """(seed, offset) for the Philox streams of the sampling kernels: one fresh offset per call."""

    def __init__(self):
        self.seed, self.offset = 0, 0

    def next(self):
        self.offset += 1
        return self.seed, self.offset

    def reserve(self, n: int) -> int:
        """Set ``n`` offsets aside and return the first one: the env-group lanes of a rollout (pipeline.py) draw
        ``first + step * groups + group`` so that their streams do not depend on which lane's thread runs first."""
        first = self.offset + 1
        self.offset += int(n)
        return first


def fused_mlp_ptrs(agent):
    """``(actor MlpNetPtrs, critic MlpNetPtrs)`` of an agent whose two networks the fused MLP kernels cover (64-64 tanh,
    obs_dim <= 512, n_out <= 20, f32 on a HIP device), else None.  Rebuilt when the parameters moved (``.to()``, flat buffers)."""
    import os

    if os.environ.get("MI355PPO_MLP", "fused") == "torch":       # A/B: keep the networks on library GEMMs
        return None
    actor, critic = agent.mlp_nets()
    w = actor[0].weight
    if not w.is_cuda or w.dtype != torch.float32:
        return None
    key = tuple(p.data_ptr() for p in list(actor.parameters()) + list(critic.parameters()))
    if agent._fused is None or agent._fused[0] != key:
        try:
            a, c = ops.MlpNetPtrs(actor), ops.MlpNetPtrs(critic)
        except AssertionError:
            agent._fused = (key, None)
            return None
        ok = ops.mlp_supported(a.obs_dim, a.n_out) and c.n_out == 1 and c.obs_dim == a.obs_dim
        agent._fused = (key, (a, c) if ok else None)
    return agent._fused[1]


def _fused_mlp(agent, x):
    """The fused forward applies when no autograd graph is being recorded (rollout, bootstrap value) on a HIP device."""
    if not x.is_cuda or x.dim() != 2 or x.dtype != torch.float32:
        return None
    if torch.is_grad_enabled() and any(p.requires_grad for p in agent.parameters()):
        return None
    return fused_mlp_ptrs(agent)


class _DiscreteMixin:
    discrete = True

    def _dist(self, logits, action):
        if logits.is_cuda:
            if action is None:
                seed, off = self.rng.next()
                a64, _, lp, ent = ops.categorical_sample(logits.contiguous(), seed=seed, offset=off)
                return a64, lp, ent
            if torch.is_grad_enabled() and logits.requires_grad:
                lp, ent = ops.CategoricalLogProbEntropy.apply(logits, action)
            else:
                lp, ent = ops.categorical_logprob_entropy(logits.contiguous(), action.contiguous())
            return action, lp, ent
        probs = Categorical(logits=logits)
        if action is None:
            action = probs.sample()
        return action, probs.log_prob(action), probs.entropy()


class AtariAgent(_DiscreteMixin, nn.Module):
    obs_is_image = True

    def __init__(self, envs):
        super().__init__()
        self.network = nn.Sequential(
            layer_init(nn.Conv2d(4, 32, 8, stride=4)),
            nn.ReLU(),
            layer_init(nn.Conv2d(32, 64, 4, stride=2)),
            nn.ReLU(),
            layer_init(nn.Conv2d(64, 64, 3, stride=1)),
            nn.ReLU(),
            nn.Flatten(),
            layer_init(nn.Linear(64 * 7 * 7, 512)),
            nn.ReLU(),
        )
        self.actor = layer_init(nn.Linear(512, envs.single_action_space.n), std=0.01)
        self.critic = layer_init(nn.Linear(512, 1), std=1)
        self.n_actions = envs.single_action_space.n
        self.rng = _SampleCounter()
        self._trunk = None

    def _normalise(self, x):
        if x.dtype == torch.uint8:
            return ops.obs_u8_to_f32(x.contiguous()) if x.is_cuda else x.float() / 255.0
        return x / 255.0

    def heads(self, xn):
        """xn: already-normalised f32 observations -> (logits (B,A), value (B,1))."""
        hidden = self.network(xn)
        return self.actor(hidden), self.critic(hidden)

    def heads_u8(self, obs_rows, inds=None):
        """The learner's fast path: ``obs_rows`` is the uint8 rollout buffer in its pixel-interleaved layout
        (rows, 84, 84, 4) and ``inds`` optionally gathers rows (``b_obs[mb_inds]``).  Same function as
        ``heads(obs / 255.0)`` with the three convolutions on the libmi355ppo f32-MFMA kernels (gather, /255, bias,
        ReLU and the ReLU / bias backward fused; cleanrl_amd/cnn.py); the two Linear layers stay on hipBLASLt."""
        from . import cnn

        if self._trunk is None:
            self._trunk = cnn.NatureTrunk()
        net = self.network
        self._trunk.bufs.pack_params = (net[0].weight, net[2].weight, net[4].weight, net[7].weight)
        feats = self._trunk(obs_rows, inds, net[0], net[2], net[4])
        hidden = cnn.LinearReLUHwcFn.apply(feats, net[7].weight, net[7].bias, self._trunk.bufs)
        if cnn.heads_supported(self.actor, self.critic):
            return cnn.HeadsFn.apply(hidden, self.actor.weight, self.actor.bias, self.critic.weight, self.critic.bias, self._trunk.bufs)
        return self.actor(hidden), self.critic(hidden)      # > 18 actions: library GEMMs

    def act_u8(self, obs_rows, seed, offset, offset_base=None, action_f32_out=None, logprob_out=None, value_out=None, want_i64=True):
        """The learner's rollout step on uint8 rows (no autograd graph): trunk, then Linear(3136,512) + heads + Categorical draw in two
        launches (``cnn.fc_heads_act_categorical``).  None when the fused step does not apply (wide action spaces, huge batches)."""
        from . import cnn

        if self._trunk is None:
            self._trunk = cnn.NatureTrunk()
        net = self.network
        if torch.is_grad_enabled() or not cnn.heads_supported(self.actor, self.critic) or not cnn.fc_heads_act_supported_rows(obs_rows.shape[0]):
            return None                                   # (decided before the trunk runs: the caller's fallback computes it)
        self._trunk.bufs.pack_params = (net[0].weight, net[2].weight, net[4].weight, net[7].weight)
        feats = self._trunk(obs_rows, None, net[0], net[2], net[4])
        if not cnn.fc_heads_act_supported(feats):
            return None
        bufs = self._trunk.bufs
        return cnn.fc_heads_act_categorical(feats, bufs.fc_pack_fwd(net[7].weight), net[7].bias.detach().contiguous(),
                                            self.actor.weight.detach(), self.actor.bias.detach(), self.critic.weight.detach(),
                                            self.critic.bias.detach(), seed, offset, offset_base, action_f32_out, logprob_out, value_out,
                                            want_i64=want_i64, amax=bufs.rec_of(cnn.REC_A3, feats) if bufs.f16(feats) else None)

    def get_value(self, x):
        return self.critic(self.network(self._normalise(x)))

    def get_action_and_value(self, x, action=None):
        logits, value = self.heads(self._normalise(x))
        action, lp, ent = self._dist(logits, action)
        return action, lp, ent, value


class MlpAgent(_DiscreteMixin, nn.Module):
    obs_is_image = False

    def __init__(self, envs):
        super().__init__()
        obs_dim = int(np.array(envs.single_observation_space.shape).prod())
        self.critic = nn.Sequential(
            layer_init(nn.Linear(obs_dim, 64)),
            nn.Tanh(),
            layer_init(nn.Linear(64, 64)),
            nn.Tanh(),
            layer_init(nn.Linear(64, 1), std=1.0),
        )
        self.actor = nn.Sequential(
            layer_init(nn.Linear(obs_dim, 64)),
            nn.Tanh(),
            layer_init(nn.Linear(64, 64)),
            nn.Tanh(),
            layer_init(nn.Linear(64, envs.single_action_space.n), std=0.01),
        )
        self.n_actions = envs.single_action_space.n
        self.rng = _SampleCounter()
        self._fused = None

    def mlp_nets(self):
        """(actor, critic) ``nn.Sequential`` pair: the seam of the fused MLP kernels (csrc/mlp.hip)."""
        return self.actor, self.critic

    def heads(self, x):
        f = _fused_mlp(self, x)
        if f is not None:                      # no autograd graph wanted: both networks in one launch
            logits, value = ops.mlp_forward(x.contiguous(), *f)
            return logits, value.unsqueeze(1)
        return self.actor(x), self.critic(x)

    def get_value(self, x):
        f = _fused_mlp(self, x)
        if f is not None:
            return ops.mlp_forward(x.contiguous(), *f)[1].unsqueeze(1)
        return self.critic(x)

    def get_action_and_value(self, x, action=None):
        f = _fused_mlp(self, x) if action is None else None
        if f is not None:                      # rollout step: forwards + sample + log_prob + entropy in one launch
            seed, off = self.rng.next()
            a64, _, lp, ent, value, _ = ops.mlp_act_categorical(x.contiguous(), *f, seed=seed, offset=off, want_entropy=True)
            return a64, lp, ent, value.unsqueeze(1)
        logits, value = self.heads(x)
        action, lp, ent = self._dist(logits, action)
        return action, lp, ent, value


class ContinuousAgent(nn.Module):
    obs_is_image = False
    discrete = False

    def __init__(self, envs, rpo_alpha=None):
        super().__init__()
        self.rpo_alpha = rpo_alpha      # rpo_continuous_action.py:109-111: perturb the mean when re-evaluating actions
        obs_dim = int(np.array(envs.single_observation_space.shape).prod())
        act_dim = int(np.prod(envs.single_action_space.shape))
        self.critic = nn.Sequential(
            layer_init(nn.Linear(obs_dim, 64)),
            nn.Tanh(),
            layer_init(nn.Linear(64, 64)),
            nn.Tanh(),
            layer_init(nn.Linear(64, 1), std=1.0),
        )
        self.actor_mean = nn.Sequential(
            layer_init(nn.Linear(obs_dim, 64)),
            nn.Tanh(),
            layer_init(nn.Linear(64, 64)),
            nn.Tanh(),
            layer_init(nn.Linear(64, act_dim), std=0.01),
        )
        self.actor_logstd = nn.Parameter(torch.zeros(1, act_dim))
        self.act_dim = act_dim
        self.rng = _SampleCounter()
        self._fused = None

    def mlp_nets(self):
        """(actor_mean, critic) ``nn.Sequential`` pair: the seam of the fused MLP kernels (csrc/mlp.hip)."""
        return self.actor_mean, self.critic

    def heads(self, x):
        f = _fused_mlp(self, x)
        if f is not None:
            mean, value = ops.mlp_forward(x.contiguous(), *f)
            return mean, value.unsqueeze(1)
        return self.actor_mean(x), self.critic(x)

    def get_value(self, x):
        f = _fused_mlp(self, x)
        if f is not None:
            return ops.mlp_forward(x.contiguous(), *f)[1].unsqueeze(1)
        return self.critic(x)

    def perturb_mean(self, mean):
        """rpo_continuous_action.py:138-142: ``action_mean + U(-alpha, alpha)`` (only when actions are re-evaluated)."""
        if self.rpo_alpha is None:
            return mean
        return mean + torch.empty_like(mean).uniform_(-self.rpo_alpha, self.rpo_alpha)

    def get_action_and_value(self, x, action=None):
        f = _fused_mlp(self, x) if action is None else None
        if f is not None:                      # rollout step: forwards + sample + log_prob + entropy in one launch
            seed, off = self.rng.next()
            act, lp, ent, value, _ = ops.mlp_act_normal(x.contiguous(), *f, self.actor_logstd.detach(), seed=seed, offset=off,
                                                        want_entropy=True)
            return act, lp, ent, value.unsqueeze(1)
        mean, value = self.heads(x)
        if action is not None:
            mean = self.perturb_mean(mean)
        if mean.is_cuda:
            if action is None:
                seed, off = self.rng.next()
                action, lp, ent = ops.normal_sample(mean.contiguous(), self.actor_logstd, seed=seed, offset=off)
            elif torch.is_grad_enabled() and (mean.requires_grad or self.actor_logstd.requires_grad):
                lp, ent = ops.NormalLogProbEntropy.apply(mean, self.actor_logstd, action)
            else:
                lp, ent = ops.normal_logprob_entropy(mean.contiguous(), self.actor_logstd, action.contiguous())
            return action, lp, ent, value
        action_logstd = self.actor_logstd.expand_as(mean)
        probs = Normal(mean, torch.exp(action_logstd))
        if action is None:
            action = probs.sample()
        return action, probs.log_prob(action).sum(1), probs.entropy().sum(1), value


class AtariLSTMAgent(_DiscreteMixin, nn.Module):
    """ppo_atari_lstm.py:117-165: NatureCNN on ONE 84x84 frame -> Linear(3136,512) -> LSTM(512,128) -> actor / critic.
    Same construction order as the reference (network layers, ``nn.LSTM`` default init, then bias := 0 and orthogonal
    weights in ``named_parameters`` order, then actor, critic), hence the same weights for the same torch seed.
    The recurrent state is reset where ``done`` is 1, one time step at a time, exactly as ``get_states`` (:138-156)."""

    obs_is_image = True
    recurrent = True

    def __init__(self, envs):
        super().__init__()
        self.network = nn.Sequential(
            layer_init(nn.Conv2d(1, 32, 8, stride=4)),
            nn.ReLU(),
            layer_init(nn.Conv2d(32, 64, 4, stride=2)),
            nn.ReLU(),
            layer_init(nn.Conv2d(64, 64, 3, stride=1)),
            nn.ReLU(),
            nn.Flatten(),
            layer_init(nn.Linear(64 * 7 * 7, 512)),
            nn.ReLU(),
        )
        self.lstm = nn.LSTM(512, 128)
        for name, param in self.lstm.named_parameters():
            if "bias" in name:
                nn.init.constant_(param, 0)
            elif "weight" in name:
                nn.init.orthogonal_(param, 1.0)
        self.actor = layer_init(nn.Linear(128, envs.single_action_space.n), std=0.01)
        self.critic = layer_init(nn.Linear(128, 1), std=1)
        self.n_actions = envs.single_action_space.n
        self.rng = _SampleCounter()

    def _normalise(self, x):
        if x.dtype == torch.uint8:
            return ops.obs_u8_to_f32(x.contiguous()) if x.is_cuda else x.float() / 255.0
        return x / 255.0

    def initial_state(self, num_envs: int, device):
        """(:225-228) zero hidden and cell state, (num_layers, N, hidden)."""
        shape = (self.lstm.num_layers, num_envs, self.lstm.hidden_size)
        return torch.zeros(shape, device=device), torch.zeros(shape, device=device)

    def states_from_features(self, hidden, lstm_state, done):
        """The LSTM logic of ``get_states`` (:141-156) on the (T*B, 512) features of T time-major steps of B envs."""
        batch_size = lstm_state[0].shape[1]
        hidden = hidden.reshape((-1, batch_size, self.lstm.input_size))
        done = done.reshape((-1, batch_size))
        new_hidden = []
        for h, d in zip(hidden, done):
            keep = (1.0 - d).view(1, -1, 1)
            h, lstm_state = self.lstm(h.unsqueeze(0), (keep * lstm_state[0], keep * lstm_state[1]))
            new_hidden += [h]
        return torch.flatten(torch.cat(new_hidden), 0, 1), lstm_state

    def heads_seq(self, xn, lstm_state, done):
        """xn: already-normalised f32 frames, time-major (T*B, 1, 84, 84) -> (logits, value, new state): the seam the
        learner's update uses (the fused loss kernel consumes logits / value; autograd runs through this function)."""
        hidden, lstm_state = self.states_from_features(self.network(xn), lstm_state, done)
        return self.actor(hidden), self.critic(hidden), lstm_state

    def get_states(self, x, lstm_state, done):
        return self.states_from_features(self.network(self._normalise(x)), lstm_state, done)

    def get_value(self, x, lstm_state, done):
        hidden, _ = self.get_states(x, lstm_state, done)
        return self.critic(hidden)

    def get_action_and_value(self, x, lstm_state, done, action=None):
        hidden, lstm_state = self.get_states(x, lstm_state, done)
        logits = self.actor(hidden)
        action, lp, ent = self._dist(logits, action)
        return action, lp, ent, self.critic(hidden), lstm_state


class ResidualBlock(nn.Module):
    """ppo_procgen.py:86-98 (IMPALA-CNN residual block): x + conv1(relu(conv0(relu(x))))."""

    def __init__(self, channels):
        super().__init__()
        self.conv0 = nn.Conv2d(in_channels=channels, out_channels=channels, kernel_size=3, padding=1)
        self.conv1 = nn.Conv2d(in_channels=channels, out_channels=channels, kernel_size=3, padding=1)

    def forward(self, x):
        inputs = x
        x = nn.functional.relu(x)
        x = self.conv0(x)
        x = nn.functional.relu(x)
        x = self.conv1(x)
        return x + inputs


class ConvSequence(nn.Module):
    """ppo_procgen.py:101-124: conv3x3 -> max_pool(3, stride 2, pad 1) -> two residual blocks."""

    def __init__(self, input_shape, out_channels):
        super().__init__()
        self._input_shape = input_shape
        self._out_channels = out_channels
        self.conv = nn.Conv2d(in_channels=self._input_shape[0], out_channels=self._out_channels, kernel_size=3, padding=1)
        self.res_block0 = ResidualBlock(self._out_channels)
        self.res_block1 = ResidualBlock(self._out_channels)

    def forward(self, x):
        x = self.conv(x)
        x = nn.functional.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        x = self.res_block0(x)
        x = self.res_block1(x)
        assert x.shape[1:] == self.get_output_shape()
        return x

    def get_output_shape(self):
        _c, h, w = self._input_shape
        return (self._out_channels, (h + 1) // 2, (w + 1) // 2)


class ProcgenAgent(_DiscreteMixin, nn.Module):
    """ppo_procgen.py:127-158: IMPALA-CNN [16, 32, 32] on (64, 64, 3) frames that arrive pixel-interleaved ("bhwc") --
    which is how the rollout buffer stores images anyway, so no relayout kernel runs for this script.  The conv layers
    keep torch's default initialisation, as in the reference (only actor / critic use ``layer_init``)."""

    obs_is_image = True
    obs_layout = "hwc"

    def __init__(self, envs):
        super().__init__()
        h, w, c = envs.single_observation_space.shape
        shape = (c, h, w)
        conv_seqs = []
        for out_channels in [16, 32, 32]:
            conv_seq = ConvSequence(shape, out_channels)
            shape = conv_seq.get_output_shape()
            conv_seqs.append(conv_seq)
        conv_seqs += [
            nn.Flatten(),
            nn.ReLU(),
            nn.Linear(in_features=shape[0] * shape[1] * shape[2], out_features=256),
            nn.ReLU(),
        ]
        self.network = nn.Sequential(*conv_seqs)
        self.actor = layer_init(nn.Linear(256, envs.single_action_space.n), std=0.01)
        self.critic = layer_init(nn.Linear(256, 1), std=1)
        self.n_actions = envs.single_action_space.n
        self.rng = _SampleCounter()

    def _normalise(self, x):
        """(B, H, W, C) frames -> normalised (B, C, H, W) view ("bhwc" -> "bchw", :147,150)."""
        if x.dtype == torch.uint8:
            x = ops.obs_u8_to_f32(x.contiguous()) if x.is_cuda else x.float() / 255.0
        else:
            x = x / 255.0
        return x.permute((0, 3, 1, 2))

    def heads(self, xn):
        """xn: normalised f32 frames as (B, C, H, W) (any strides) -> (logits, value)."""
        hidden = self.network(xn)
        return self.actor(hidden), self.critic(hidden)

    def get_value(self, x):
        return self.critic(self.network(self._normalise(x)))

    def get_action_and_value(self, x, action=None):
        logits, value = self.heads(self._normalise(x))
        action, lp, ent = self._dist(logits, action)
        return action, lp, ent, value


class RNDAgent(_DiscreteMixin, nn.Module):
    """ppo_rnd_envpool.py:139-185: NatureCNN -> Linear(3136,256) -> Linear(256,448), a 448-448 "extra" layer feeding two
    value heads (extrinsic / intrinsic) through a residual sum, and a two-layer actor.  Same construction order as the
    reference."""

    obs_is_image = True
    two_value_heads = True

    def __init__(self, envs):
        super().__init__()
        self.network = nn.Sequential(
            layer_init(nn.Conv2d(4, 32, 8, stride=4)),
            nn.ReLU(),
            layer_init(nn.Conv2d(32, 64, 4, stride=2)),
            nn.ReLU(),
            layer_init(nn.Conv2d(64, 64, 3, stride=1)),
            nn.ReLU(),
            nn.Flatten(),
            layer_init(nn.Linear(64 * 7 * 7, 256)),
            nn.ReLU(),
            layer_init(nn.Linear(256, 448)),
            nn.ReLU(),
        )
        self.extra_layer = nn.Sequential(layer_init(nn.Linear(448, 448), std=0.1), nn.ReLU())
        self.actor = nn.Sequential(
            layer_init(nn.Linear(448, 448), std=0.01),
            nn.ReLU(),
            layer_init(nn.Linear(448, envs.single_action_space.n), std=0.01),
        )
        self.critic_ext = layer_init(nn.Linear(448, 1), std=0.01)
        self.critic_int = layer_init(nn.Linear(448, 1), std=0.01)
        self.n_actions = envs.single_action_space.n
        self.rng = _SampleCounter()

    def _normalise(self, x):
        if x.dtype == torch.uint8:
            return ops.obs_u8_to_f32(x.contiguous()) if x.is_cuda else x.float() / 255.0
        return x / 255.0

    def heads3(self, xn):
        """xn: normalised f32 observations (B,4,84,84) -> (logits, extrinsic value, intrinsic value)."""
        hidden = self.network(xn)
        logits = self.actor(hidden)
        features = self.extra_layer(hidden)
        return logits, self.critic_ext(features + hidden), self.critic_int(features + hidden)

    def get_action_and_value(self, x, action=None):
        logits, v_ext, v_int = self.heads3(self._normalise(x))
        action, lp, ent = self._dist(logits, action)
        return action, lp, ent, v_ext, v_int

    def get_value(self, x):
        _, v_ext, v_int = self.heads3(self._normalise(x))
        return v_ext, v_int


class RNDModel(nn.Module):
    """ppo_rnd_envpool.py:188-233: predictor and frozen random target network on ONE normalised 84x84 frame."""

    def __init__(self, input_size, output_size):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        feature_output = 7 * 7 * 64
        self.predictor = nn.Sequential(
            layer_init(nn.Conv2d(in_channels=1, out_channels=32, kernel_size=8, stride=4)),
            nn.LeakyReLU(),
            layer_init(nn.Conv2d(in_channels=32, out_channels=64, kernel_size=4, stride=2)),
            nn.LeakyReLU(),
            layer_init(nn.Conv2d(in_channels=64, out_channels=64, kernel_size=3, stride=1)),
            nn.LeakyReLU(),
            nn.Flatten(),
            layer_init(nn.Linear(feature_output, 512)),
            nn.ReLU(),
            layer_init(nn.Linear(512, 512)),
            nn.ReLU(),
            layer_init(nn.Linear(512, 512)),
        )
        self.target = nn.Sequential(
            layer_init(nn.Conv2d(in_channels=1, out_channels=32, kernel_size=8, stride=4)),
            nn.LeakyReLU(),
            layer_init(nn.Conv2d(in_channels=32, out_channels=64, kernel_size=4, stride=2)),
            nn.LeakyReLU(),
            layer_init(nn.Conv2d(in_channels=64, out_channels=64, kernel_size=3, stride=1)),
            nn.LeakyReLU(),
            nn.Flatten(),
            layer_init(nn.Linear(feature_output, 512)),
        )
        for param in self.target.parameters():          # target network is not trainable
            param.requires_grad = False

    def forward(self, next_obs):
        target_feature = self.target(next_obs)
        predict_feature = self.predictor(next_obs)
        return predict_feature, target_feature


def layer_init_normed(layer, norm_dim, scale=1.0):
    """ppg_procgen.py:100-104: rescale every output unit's weight vector to norm ``scale``; zero bias."""
    with torch.no_grad():
        layer.weight.data *= scale / layer.weight.norm(dim=norm_dim, p=2, keepdim=True)
        layer.bias *= 0
    return layer


class ResidualBlockNormed(nn.Module):
    """ppg_procgen.py:123-139."""

    def __init__(self, channels, scale):
        super().__init__()
        scale = np.sqrt(scale)
        conv0 = nn.Conv2d(in_channels=channels, out_channels=channels, kernel_size=3, padding=1)
        self.conv0 = layer_init_normed(conv0, norm_dim=(1, 2, 3), scale=scale)
        conv1 = nn.Conv2d(in_channels=channels, out_channels=channels, kernel_size=3, padding=1)
        self.conv1 = layer_init_normed(conv1, norm_dim=(1, 2, 3), scale=scale)

    def forward(self, x):
        inputs = x
        x = nn.functional.relu(x)
        x = self.conv0(x)
        x = nn.functional.relu(x)
        x = self.conv1(x)
        return x + inputs


class ConvSequenceNormed(nn.Module):
    """ppg_procgen.py:142-165."""

    def __init__(self, input_shape, out_channels, scale):
        super().__init__()
        self._input_shape = input_shape
        self._out_channels = out_channels
        conv = nn.Conv2d(in_channels=self._input_shape[0], out_channels=self._out_channels, kernel_size=3, padding=1)
        self.conv = layer_init_normed(conv, norm_dim=(1, 2, 3), scale=1.0)
        nblocks = 2
        scale = scale / np.sqrt(nblocks)
        self.res_block0 = ResidualBlockNormed(self._out_channels, scale=scale)
        self.res_block1 = ResidualBlockNormed(self._out_channels, scale=scale)

    def forward(self, x):
        x = self.conv(x)
        x = nn.functional.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        x = self.res_block0(x)
        x = self.res_block1(x)
        assert x.shape[1:] == self.get_output_shape()
        return x

    def get_output_shape(self):
        _c, h, w = self._input_shape
        return (self._out_channels, (h + 1) // 2, (w + 1) // 2)


class PPGAgent(_DiscreteMixin, nn.Module):
    """ppg_procgen.py:168-211: IMPALA-CNN with norm-scaled initialisation, a policy head, a value head on the DETACHED
    features (the policy phase's value loss does not shape the encoder) and an auxiliary value head on the live features
    (the auxiliary phase's does)."""

    obs_is_image = True
    obs_layout = "hwc"

    def __init__(self, envs):
        super().__init__()
        h, w, c = envs.single_observation_space.shape
        shape = (c, h, w)
        conv_seqs = []
        chans = [16, 32, 32]
        scale = 1 / np.sqrt(len(chans))
        for out_channels in chans:
            conv_seq = ConvSequenceNormed(shape, out_channels, scale=scale)
            shape = conv_seq.get_output_shape()
            conv_seqs.append(conv_seq)
        encodertop = nn.Linear(in_features=shape[0] * shape[1] * shape[2], out_features=256)
        encodertop = layer_init_normed(encodertop, norm_dim=1, scale=1.4)
        conv_seqs += [
            nn.Flatten(),
            nn.ReLU(),
            encodertop,
            nn.ReLU(),
        ]
        self.network = nn.Sequential(*conv_seqs)
        self.actor = layer_init_normed(nn.Linear(256, envs.single_action_space.n), norm_dim=1, scale=0.1)
        self.critic = layer_init_normed(nn.Linear(256, 1), norm_dim=1, scale=0.1)
        self.aux_critic = layer_init_normed(nn.Linear(256, 1), norm_dim=1, scale=0.1)
        self.n_actions = envs.single_action_space.n
        self.rng = _SampleCounter()

    def _normalise(self, x):
        """(B, H, W, C) frames -> normalised (B, C, H, W) view ("bhwc" -> "bchw")."""
        if x.dtype == torch.uint8:
            x = ops.obs_u8_to_f32(x.contiguous()) if x.is_cuda else x.float() / 255.0
        else:
            x = x / 255.0
        return x.permute((0, 3, 1, 2))

    def heads(self, xn):
        """xn: normalised (B, C, H, W) frames -> (logits, value on the detached features): the policy-phase seam."""
        hidden = self.network(xn)
        return self.actor(hidden), self.critic(hidden.detach())

    def heads_aux(self, xn):
        """-> (logits, value on detached features, auxiliary value on live features): the auxiliary-phase seam."""
        hidden = self.network(xn)
        return self.actor(hidden), self.critic(hidden.detach()), self.aux_critic(hidden)

    def get_action_and_value(self, x, action=None):
        logits, value = self.heads(self._normalise(x))
        action, lp, ent = self._dist(logits, action)
        return action, lp, ent, value

    def get_value(self, x):
        return self.critic(self.network(self._normalise(x)))

    def get_pi_value_and_aux_value(self, x):
        logits, value, aux = self.heads_aux(self._normalise(x))
        return Categorical(logits=logits), value, aux

    def get_pi(self, x):
        return Categorical(logits=self.actor(self.network(self._normalise(x))))


class MAAtariAgent(_DiscreteMixin, nn.Module):
    """ppo_pettingzoo_ma_atari.py:86-118: NatureCNN on (84, 84, 6) pixel-interleaved observations -- four stacked frames plus
    two agent-indicator planes.  Only the four frame channels are divided by 255 (:104,109)."""

    obs_is_image = True
    obs_layout = "hwc"
    frame_channels = 4            # channels [0, 4) are pixels; the rest pass through unscaled

    def __init__(self, envs):
        super().__init__()
        self.network = nn.Sequential(
            layer_init(nn.Conv2d(6, 32, 8, stride=4)),
            nn.ReLU(),
            layer_init(nn.Conv2d(32, 64, 4, stride=2)),
            nn.ReLU(),
            layer_init(nn.Conv2d(64, 64, 3, stride=1)),
            nn.ReLU(),
            nn.Flatten(),
            layer_init(nn.Linear(64 * 7 * 7, 512)),
            nn.ReLU(),
        )
        self.actor = layer_init(nn.Linear(512, envs.single_action_space.n), std=0.01)
        self.critic = layer_init(nn.Linear(512, 1), std=1)
        self.n_actions = envs.single_action_space.n
        self.rng = _SampleCounter()

    def scale_frames_(self, x):
        """In place on a (B, H, W, C) f32 tensor holding raw 0..255 values: ``x[:, :, :, [0,1,2,3]] /= 255.0``."""
        x[..., : self.frame_channels] /= 255.0
        return x

    def _normalise(self, x):
        if x.dtype == torch.uint8:
            x = ops.obs_u8_to_f32(x.contiguous(), scale_255=False) if x.is_cuda else x.float()
        else:
            x = x.clone()
        return self.scale_frames_(x).permute((0, 3, 1, 2))

    def heads(self, xn):
        """xn: normalised (B, 6, 84, 84) -> (logits, value)."""
        hidden = self.network(xn)
        return self.actor(hidden), self.critic(hidden)

    def get_value(self, x):
        return self.critic(self.network(self._normalise(x)))

    def get_action_and_value(self, x, action=None):
        logits, value = self.heads(self._normalise(x))
        action, lp, ent = self._dist(logits, action)
        return action, lp, ent, value
