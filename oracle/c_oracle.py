"""ctypes wrapper of the scalar C oracle ``oracle/c/ppo_oracle.c`` (TEST INFRASTRUCTURE)."""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_double, c_int, c_int64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libppo_oracle.so")
_lib = None


def build() -> str:
    subprocess.run(["make", "-C", os.path.join(_HERE, "c")], check=True, capture_output=True)
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        P = c_void_p
        _lib.oracle_gae_f32.argtypes = [P, P, P, P, P, P, P, c_int, c_int, c_double, c_double]
        _lib.oracle_categorical_sample_f32.argtypes = [P, P, P, P, P, c_int, c_int]
        _lib.oracle_loss_categorical_f32.argtypes = [P, P, P, P, P, P, P, P, c_int, c_int, c_double, c_double, c_double,
                                                     c_int, c_int, P, P, P]
        _lib.oracle_obs_u8_to_f32.argtypes = [P, P, P, c_int64, c_int64, c_int]
        for f in (_lib.oracle_gae_f32, _lib.oracle_categorical_sample_f32, _lib.oracle_loss_categorical_f32,
                  _lib.oracle_obs_u8_to_f32):
            f.restype = None
    return _lib


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _p(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def gae(rewards, dones, values, next_done, next_value, gamma, gae_lambda):
    r, d, v = _c(rewards, np.float32), _c(dones, np.float32), _c(values, np.float32)
    nd, nv = _c(next_done, np.float32).reshape(-1), _c(next_value, np.float32).reshape(-1)
    T, N = r.shape
    adv, ret = np.empty_like(r), np.empty_like(r)
    load().oracle_gae_f32(_p(r), _p(d), _p(v), _p(nd), _p(nv), _p(adv), _p(ret), T, N, float(gamma), float(gae_lambda))
    return adv, ret


def categorical_sample(logits, noise_exp1):
    x, q = _c(logits, np.float32), _c(noise_exp1, np.float32)
    B, A = x.shape
    act = np.empty(B, np.int64)
    lp, ent = np.empty(B, np.float32), np.empty(B, np.float32)
    load().oracle_categorical_sample_f32(_p(x), _p(q), _p(act), _p(lp), _p(ent), B, A)
    return act, lp, ent


def loss_categorical(logits, value, mb_inds, b_actions, b_logprobs, b_adv, b_ret, b_val, clip_coef, ent_coef, vf_coef,
                     norm_adv, clip_vloss):
    x, v = _c(logits, np.float32), _c(value, np.float32).reshape(-1)
    M, A = x.shape
    inds = None if mb_inds is None else _c(mb_inds, np.int64)
    ba, bl, bad, br, bv = (_c(t, np.float32).reshape(-1) for t in (b_actions, b_logprobs, b_adv, b_ret, b_val))
    sc, dl, dv = np.empty(7, np.float32), np.empty_like(x), np.empty(M, np.float32)
    load().oracle_loss_categorical_f32(_p(x), _p(v), _p(inds), _p(ba), _p(bl), _p(bad), _p(br), _p(bv), M, A,
                                       float(clip_coef), float(ent_coef), float(vf_coef), int(norm_adv), int(clip_vloss),
                                       _p(sc), _p(dl), _p(dv))
    return sc, dl, dv


def obs_u8_to_f32(src, inds, scale_255=True):
    s = _c(src, np.uint8)
    rb = int(np.prod(s.shape[1:]))
    ii = None if inds is None else _c(inds, np.int64)
    rows = s.shape[0] if ii is None else ii.shape[0]
    out = np.empty((rows,) + s.shape[1:], np.float32)
    load().oracle_obs_u8_to_f32(_p(s), _p(ii), _p(out), rows, rb, int(scale_255))
    return out
