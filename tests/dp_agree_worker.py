"""Worker of tests/test_host_logic.py::test_all_ranks_agree...: one rank of a gloo group on the CPU; rank `bad` (argv[2], -1: none) reports a
failed capture.  Prints `rank R agreed=<bool> policy=<str>`.  Not a test module."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleanrl_amd.learner import all_ranks_agree, update_graph_policy  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
bad = int(sys.argv[1])
print(f"rank {rank} agreed={all_ranks_agree(rank != bad, torch.device('cpu'))} policy={update_graph_policy(world)}", flush=True)
dist.barrier()
dist.destroy_process_group()
