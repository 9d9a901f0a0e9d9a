"""Worker of ``test_gpu_multirank.py::test_update_graphs_with_two_ranks...``: ONE rank of a data-parallel run with the update
replayed from captured graphs -- per (epoch, minibatch) slot three hipGraphs with the gradient exchange between them
(``learner._SlotGraphs``) -- beside an eager twin from the same seeds, both ranks on ``cuda:0`` over gloo.  The two learners of
a rank issue the same collectives in the same order, so the twins interleave safely.  Reference: ppo_atari_multigpu.py:314-377.
Not a test module."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from cleanrl_amd import envs as E, learner_smoke  # noqa: E402
from cleanrl_amd.agents import AtariAgent  # noqa: E402
from cleanrl_amd.learner import PPOLearner  # noqa: E402


def main(out_dir, N, T, nmb, epochs, iters, no_early=False, peer=False):
    rank, world = int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if peer:             # the flat gradient over HIP IPC segments (csrc/dpcomm.hip): gloo carries the handles and the agreement only; a slot is ONE graph
        os.environ["MI355PPO_ALLREDUCE"] = "peer"
    if no_early:         # the arrangement the policy chooses over RCCL: no early bucket -> one all-reduce behind the backward, two graphs per slot
        import cleanrl_amd.learner as learner_mod

        learner_mod.early_bucket_policy = lambda world_size: False
        learner_mod.update_graph_policy = lambda world_size: "capture+check"      # ... and the route bench.py / runner.train take there: capture_update_agreed

    def make(graphs):
        torch.manual_seed(4)                                    # same init on every rank (ppo_atari_multigpu.py:211)
        env = E.DeviceSyntheticAtariVecEnv(N, dev, seed=6 + rank, done_p=0.1)      # per-rank data (:206-212)
        agent = AtariAgent(env).to(dev)
        args = learner_smoke.default_args(num_steps=T, num_minibatches=nmb, update_epochs=epochs)
        L = PPOLearner(agent, args, env.single_observation_space, env.single_action_space, N, dev, world_size=world, sample_seed=8 + rank)
        L.observe(0, env.obs_into(L.stage_obs), L.dones[0])
        assert (L._peer is not None) == peer
        if graphs and (no_early or peer):
            assert L.capture_update_agreed(), "capture, self-check and the all-ranks agreement (the RCCL policy's route, here over gloo)"
        elif graphs:
            L.capture_update()
        return L, env

    (Le, enve), (Lg, envg) = make(False), make(True)
    segs = sorted({len(s.segs) for row in Lg._update_graphs for s in row})
    early = all(s.early == Lg._ar_early and s.early is not None for row in Lg._update_graphs for s in row)
    if peer:
        import cleanrl_amd.learner as learner_mod

        def no_pg_all_reduce(*a, **k):     # from here on the gradients must not touch the process group (barriers / agreement flags still may)
            raise AssertionError("dist.all_reduce called on the peer-memory route")
        real_all_reduce = dist.all_reduce
        learner_mod.dist.all_reduce = lambda t, *a, **k: real_all_reduce(t, *a, **k) if t.numel() <= 4 else no_pg_all_reduce()
    # the start-up check of the RCCL policy (PPOLearner.self_check_update_graphs): one captured against one eager update from the same state, state restored
    before = Lg.flat.params.clone()
    rng0 = np.random.get_state()[1][:4].copy()
    checked = bool(Lg.self_check_update_graphs())
    restored = bool(torch.equal(before, Lg.flat.params) and not Lg.flat.grads.any() and Lg.flat.step == 0 and np.array_equal(rng0, np.random.get_state()[1][:4]))
    if no_early or peer:         # (the eager twin issues the same single all-reduce: keep the twins' collectives in step)
        assert Le._ar_early is None and Lg._ar_early is None
    same = [bool(torch.equal(Le.flat.params, Lg.flat.params) and not Lg.flat.grads.any())]
    scal = []
    for it in range(iters):
        learner_smoke.rollout(Le, enve)
        learner_smoke.rollout(Lg, envg)
        np.random.seed(100 + it + 10 * rank)
        me = Le.update(2.5e-4 * (1 - it / iters))
        np.random.seed(100 + it + 10 * rank)
        mg = Lg.update(2.5e-4 * (1 - it / iters))
        Le.start_iteration(); Lg.start_iteration()
        same.append(bool(torch.equal(Le.flat.params, Lg.flat.params) and torch.equal(Le.flat.exp_avg, Lg.flat.exp_avg)
                         and torch.equal(Le.flat.exp_avg_sq, Lg.flat.exp_avg_sq)))
        scal.append(all(me[k] == mg[k] or (np.isnan(me[k]) and np.isnan(mg[k])) for k in me))
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), params=Lg.flat.params.cpu().numpy(), params_eager=Le.flat.params.cpu().numpy(),
             same=np.array(same), scalars_same=np.array(scal), segs=np.array(segs), early=early, moved=float((Lg.flat.params != 0).float().mean()),
             self_check=checked, self_check_restored=restored, peer_status_ok=bool(not peer or (Lg._peer.status() is None and Le._peer.status() is None)))
    dist.barrier()
    if peer:
        Lg._peer.close(); Le._peer.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], *(int(x) for x in sys.argv[2:7]), no_early=len(sys.argv) > 7 and sys.argv[7] == "noearly", peer=len(sys.argv) > 7 and sys.argv[7] == "peer")
