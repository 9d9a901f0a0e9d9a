"""Worker of ``test_gpu_multirank.py::test_config_d_whole_iteration_two_ranks...``: ONE rank of BASELINE configs[3] at its per-GPU
size (256 envs x 128 steps, 16 updates of 8,192 local rows), launched twice by ``torch.distributed.run`` with both ranks on
``cuda:0`` over gloo (RCCL refuses two ranks on one device).  Teacher-forced through the whole iteration of
ppo_atari_multigpu.py's own lines (tests/golden/atari_iteration_cfgD.npz); dumps what it measured.  Not a test module."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import load_golden  # noqa: E402
from whole_iteration import run_atari_iteration  # noqa: E402


def main(out_dir, graphs=False):
    rank, world = int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden("atari_iteration_cfgD")["atari_T128_N256_world2"]
    assert int(g["world_size"]) == world
    out = run_atari_iteration(g, dev, rank=rank, world=world, graphs=graphs)     # graphs: three hipGraphs per slot, collectives between them
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{k: np.asarray(v) for k, v in out.items()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], graphs=len(sys.argv) > 2 and sys.argv[2] == "graphs")
