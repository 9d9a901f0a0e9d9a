"""Run the reference's *own* source lines against synthetic tensors.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Works only where
``/root/reference`` exists, i.e. in the build container; nothing that runs on
the GPU box may import this module.

The reference scripts cannot be imported (``gymnasium``/``tyro``/``tensorboard``
are absent and the scripts are not importable modules anyway), but the hot
path's lines run unmodified under the installed torch.  We therefore

* parse a script with :mod:`ast` and compile only ``layer_init`` and ``Agent``
  (this skips the failing imports), and
* ``exec`` ``textwrap.dedent`` of the GAE block and of the minibatch-loss block,
  located by their stable marker comments, inside a namespace that provides
  ``args``, ``agent``, the rollout tensors, ...

No reference source text is stored in this repository: it is read from
``/root/reference`` at run time.
"""
from __future__ import annotations

import ast
import os
import textwrap
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim
from torch.distributions.categorical import Categorical
from torch.distributions.normal import Normal

REFERENCE_ROOT = os.environ.get("CLEANRL_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "cleanrl", "ppo.py"))


def _read(script: str) -> list[str]:
    with open(os.path.join(REFERENCE_ROOT, "cleanrl", script)) as fh:
        return fh.read().splitlines()


def _find(lines: list[str], needle: str, start: int = 0) -> int:
    for i in range(start, len(lines)):
        if needle in lines[i]:
            return i
    raise LookupError(needle)


def line_ranges(script: str) -> dict[str, tuple[int, int]]:
    """1-based inclusive line ranges of the hot-path blocks of ``script``.

    For ``ppo_atari_envpool.py`` this yields gae=(251,263), loss=(282,317),
    step=(319,322); for ``ppo_atari_multigpu.py`` gae=(288,301), loss=(320,355).
    """
    lines = _read(script)
    g0 = _find(lines, "# bootstrap value if not done") + 1  # the `with torch.no_grad():` line
    g1 = _find(lines, "returns = advantages + values", g0)
    l0 = _find(lines, "_, newlogprob, entropy, newvalue = agent.get_action_and_value", g1)
    l1 = _find(lines, "loss = pg_loss - args.ent_coef * entropy_loss", l0)
    s0 = _find(lines, "optimizer.zero_grad()", l1)
    s1 = _find(lines, "optimizer.step()", s0)
    return {"gae": (g0 + 1, g1 + 1), "loss": (l0 + 1, l1 + 1), "step": (s0 + 1, s1 + 1)}


def _block(script: str, name: str) -> str:
    lo, hi = line_ranges(script)[name]
    return textwrap.dedent("\n".join(_read(script)[lo - 1 : hi]))


def load_agent_class(script: str):
    """Compile ``layer_init`` and ``Agent`` of ``cleanrl/<script>`` verbatim."""
    src = "\n".join(_read(script))
    tree = ast.parse(src)
    names = ("layer_init", "Agent", "ResidualBlock", "ConvSequence")        # the last two: ppo_procgen.py's IMPALA-CNN blocks
    wanted = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert {n.name for n in wanted} >= {"layer_init", "Agent"}, script
    ns = {"np": np, "torch": torch, "nn": nn, "Categorical": Categorical, "Normal": Normal}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), f"<reference:{script}>", "exec"), ns)
    return ns["Agent"], ns["layer_init"]


def fake_envs(obs_shape, n_actions=None, action_shape=None):
    """Stand-in for ``envs.single_observation_space`` / ``single_action_space``."""
    if n_actions is not None:
        act = SimpleNamespace(n=int(n_actions), shape=())
    else:
        act = SimpleNamespace(shape=tuple(action_shape))
    return SimpleNamespace(
        single_observation_space=SimpleNamespace(shape=tuple(obs_shape)),
        single_action_space=act,
    )


def make_args(**kw):
    d = dict(
        gamma=0.99,
        gae_lambda=0.95,
        clip_coef=0.1,
        ent_coef=0.01,
        vf_coef=0.5,
        norm_adv=True,
        clip_vloss=True,
        max_grad_norm=0.5,
        target_kl=None,
        learning_rate=2.5e-4,
    )
    d.update(kw)
    return SimpleNamespace(**d)


class _ValueStub:
    """``agent.get_value(next_obs)`` replacement returning a preset tensor."""

    def __init__(self, next_value):
        self._v = next_value

    def get_value(self, _x):
        return self._v


def run_gae(script, rewards, dones, values, next_done, next_value, gamma, gae_lambda):
    """Execute the reference GAE block (e.g. ``ppo_atari_envpool.py:251-263``)."""
    T = rewards.shape[0]
    ns = {
        "torch": torch,
        "np": np,
        "args": SimpleNamespace(num_steps=T, gamma=gamma, gae_lambda=gae_lambda),
        "agent": _ValueStub(next_value.reshape(-1, 1)),
        "next_obs": None,
        "next_done": next_done,
        "rewards": rewards,
        "dones": dones,
        "values": values,
        "device": torch.device("cpu"),
    }
    exec(_block(script, "gae"), ns)
    return ns["advantages"], ns["returns"]


def run_loss(script, agent, args, b_obs, b_actions, b_logprobs, b_advantages, b_returns, b_values, mb_inds,
             step=False, optimizer=None):
    """Execute the reference minibatch-loss block (e.g. ``ppo_atari_envpool.py:282-317``)
    and, if ``step``, the backward/clip/Adam block (``:319-322``).

    Returns the exec namespace (``loss``, ``pg_loss``, ``v_loss``, ``entropy_loss``,
    ``old_approx_kl``, ``approx_kl``, ``clipfracs``, ``newlogprob``, ``newvalue`` ...).
    """
    ns = {
        "torch": torch,
        "np": np,
        "nn": nn,
        "args": args,
        "agent": agent,
        "optimizer": optimizer,
        "b_obs": b_obs,
        "b_actions": b_actions,
        "b_logprobs": b_logprobs,
        "b_advantages": b_advantages,
        "b_returns": b_returns,
        "b_values": b_values,
        "mb_inds": mb_inds,
        "clipfracs": [],
    }
    exec(_block(script, "loss"), ns)
    if step:
        exec(_block(script, "step"), ns)
    return ns


def make_optimizer(agent, lr):
    """``optim.Adam(agent.parameters(), lr=args.learning_rate, eps=1e-5)`` (ppo.py:168)."""
    return optim.Adam(agent.parameters(), lr=lr, eps=1e-5)
