"""cleanrl_amd -- MI355X-native PPO hot path behind CleanRL's single-file CLI surface.

The package holds only what the PPO path needs: ``csrc/`` (HIP kernels + the C ABI
``libmi355ppo.so``), the ctypes binding (``_lib``), tensor-level operators (``ops``), the
learner (``learner``), host-side environments/CLI helpers, and the drop-in scripts
``ppo.py``, ``ppo_atari.py``, ``ppo_atari_envpool.py``, ``ppo_atari_multigpu.py``,
``ppo_continuous_action.py``.
"""
__version__ = "0.1.0"
