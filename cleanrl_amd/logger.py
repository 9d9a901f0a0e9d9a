"""Scalar logging with the reference's tags (``charts/*``, ``losses/*``; docs/rl-algorithms/ppo.md:68-78).

Uses ``torch.utils.tensorboard.SummaryWriter`` when tensorboard is installed (as the reference does,
ppo.py:147-151); otherwise writes the same (tag, value, step) triples to ``runs/<run_name>/scalars.jsonl``.
"""
from __future__ import annotations

import json
import os

import numpy as np


class JsonlWriter:
    def __init__(self, log_dir: str):
        os.makedirs(log_dir, exist_ok=True)
        self.log_dir = log_dir
        self._fh = open(os.path.join(log_dir, "scalars.jsonl"), "a")

    def add_scalar(self, tag, value, step):
        value = float(np.asarray(value, dtype=np.float64).reshape(-1)[0])
        self._fh.write(json.dumps({"tag": tag, "value": value, "step": int(step)}) + "\n")

    def add_text(self, tag, text):
        with open(os.path.join(self.log_dir, tag.replace("/", "_") + ".md"), "w") as fh:
            fh.write(text)

    def flush(self):
        self._fh.flush()

    def close(self):
        self._fh.close()


def make_writer(run_name: str):
    log_dir = f"runs/{run_name}"
    try:
        from torch.utils.tensorboard import SummaryWriter

        return SummaryWriter(log_dir)
    except Exception:
        return JsonlWriter(log_dir)
