"""The world_size > 1 HIP path executed by TWO REAL PROCESSES on the one GPU a test box has (SURVEY §8 rows a9 / e).

RCCL refuses two ranks on one device, so the ranks talk over gloo, which accepts device tensors: the code under test --
``learner.py``'s ``dist.all_reduce(self.flat.grads)`` on the persistent flat buffer, ``grad_scale = 1 / world_size`` in the
fused clip + Adam kernel, the per-rank seeding protocol of the drop-in script -- is exactly what runs over RCCL on an
8-GPU node; only the transport differs.  Reference: ppo_atari_multigpu.py:166-177 (rendezvous), :206-212,231 (seeds),
:360-377 (collective block); its own test is the 2-process gloo run of tests/test_atari_multigpu.py:4-9."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(script_and_args, timeout=600, nproc=2, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--nnodes=1", "--nproc-per-node", str(nproc), "--local-addr",
           "127.0.0.1"] + script_and_args
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + "\n" + out.stderr[-3000:]
    return out.stdout + out.stderr


def _cos(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


def test_two_ranks_execute_the_collective_block_against_the_reference_golden(tmp_path):
    _torchrun([os.path.join("tests", "dp_gloo_worker.py"), str(tmp_path)])
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in (0, 1))
    assert int(r0["world"]) == int(r1["world"]) == 2 and bytes(r0["backend"]) == b"gloo"
    g = load_golden("update_step")["multigpu_cnn_world2"]
    # each rank's own loss is the reference's for its half of the data
    np.testing.assert_allclose(float(r0["loss"]), g["loss_rank0"], rtol=1e-4)
    np.testing.assert_allclose(float(r1["loss"]), g["loss_rank1"], rtol=1e-4)
    # the collective left the SAME bits in both flat gradient buffers, and the replicas are bit-identical after the step
    assert np.array_equal(r0["reduced"], r1["reduced"]), "all_reduce(SUM) results differ between the ranks"
    assert np.array_equal(r0["params"], r1["params"]), "replicas diverged after one data-parallel step"
    assert float(r0["total_norm"][0]) == float(r1["total_norm"][0])
    # pre-Adam: (g0 + g1) / world, clipped at 0.5 (:368-376), against what the reference's optimizer.step() saw
    n = float(r0["total_norm"][0])                                   # the kernel's norm of the averaged gradient
    avg = r0["reduced"] / 2.0
    np.testing.assert_allclose(np.linalg.norm(avg.astype(np.float64)), n, rtol=1e-5)
    clipped = avg * min(1.0, 0.5 / (n + 1e-6))
    s = int(g["step_grad_stride"])
    want = g["step_grad_sub"]
    # tolerance: 1e-3 of the gradient's largest element (f32 conv backward with other summation trees), cosine > 0.99999
    assert np.abs(clipped[::s] - want).max() <= 1e-3 * float(g["step_grad_absmax"])
    assert _cos(clipped[::s], want) > 0.99999
    np.testing.assert_allclose(np.linalg.norm(clipped.astype(np.float64)), float(g["step_grad_norm"]), rtol=1e-3)
    # post-Adam: the reference's parameter delta
    stride = int(g["stride"])
    delta = r0["params"][::stride] - g["init_params_sub"]
    close = np.isclose(delta, g["delta_sub"], rtol=1e-2, atol=1e-5)
    assert close.mean() > 0.99, f"only {close.mean():.4f} of sampled parameters match the reference update"


def test_config_d_whole_iteration_two_ranks_against_the_reference_lines(tmp_path):
    """BASELINE configs[3] at its per-GPU size with world = 2: each rank's 256 envs x 128 steps, 16 updates of 8,192 local rows
    with the flat gradient SUM-all-reduced across the two processes and divided by the world size in the fused clip + Adam
    kernel, against one whole iteration of ppo_atari_multigpu.py's own lines :287-377 executed by two reference ranks
    (tests/golden/atari_iteration_cfgD.npz, oracle/mint_full_size.py).  Checked per rank: rollout values, GAE, the scalars of
    all 16 local minibatches, the averaged + clipped gradient at updates 1 / 8 / 16, the parameters after update 16; and the
    replicas stay bit-identical.  Bars as config B's test (the later updates' bars are multiples of the reference's distance
    from itself under another summation order)."""
    from whole_iteration import check_atari_iteration

    _torchrun([os.path.join("tests", "dp_cfgd_worker.py"), str(tmp_path)], timeout=1500)
    g = load_golden("atari_iteration_cfgD")["atari_T128_N256_world2"]
    outs = [dict(np.load(tmp_path / f"rank{r}.npz")) for r in (0, 1)]
    assert outs[0]["params_checksum"] == outs[1]["params_checksum"] and np.array_equal(outs[0]["final_params_sub"], outs[1]["final_params_sub"]), \
        "the replicas diverged"
    for k in (1, 8, 16):
        assert np.array_equal(outs[0][f"grad{k}_sub"], outs[1][f"grad{k}_sub"]), f"all-reduced gradients differ between the ranks at update {k}"
    # (update 16's whole-vector norm: 3e-3 here, 1e-3 in configs B / C.  There the gradient is CLIPPED at updates 8 / 16 -- both sides' norm is
    #  max_grad_norm by construction --, config D's update 16 is not (reference's norm 0.196 < 0.5): the figure then measures the trajectory's
    #  own drift like the per-tensor norms do, whose reference-against-itself distance at update 16 is 1.6e-3
    #  (tests/golden/atari_iteration_cfgB_ref_sensitivity.json).  Measured: +7.1e-4 with kernel P's summation order, +1.01e-3 with kernel U's.)
    bars = {1: (1e-3, 1e-5, 1e-3, 2e-3), 8: (1e-3, 1e-5, 1e-3, 5e-3), 16: (2e-3, 5e-5, 3e-3, 1.2e-2)}
    problems, report = [], []
    for r in (0, 1):
        o = {k: (v.item() if v.ndim == 0 else v) for k, v in outs[r].items()}
        rep = []
        problems += [f"rank {r}: {p}" for p in check_atari_iteration(o, g, bars, sfx=f"_rank{r}", report=rep)]
        report += [f"rank {r}: {x}" for x in rep]
    print("\n".join(["config D whole iteration vs the reference's lines:"] + report))
    assert not problems, "\n".join(problems + report)


def test_config_d_whole_iteration_two_ranks_under_update_graphs(tmp_path):
    """The same golden with the update replayed from captured graphs (world > 1: three hipGraphs per slot with the all-reduces between
    them; bench.py's and runner.train's default route): the scalars of all 16 local minibatches and the parameters after update 16 at the
    eager test's bars (the pre-Adam gradients are inside the graphs), replicas bit-identical."""
    from whole_iteration import check_atari_iteration

    _torchrun([os.path.join("tests", "dp_cfgd_worker.py"), str(tmp_path), "graphs"], timeout=1500)
    g = load_golden("atari_iteration_cfgD")["atari_T128_N256_world2"]
    outs = [dict(np.load(tmp_path / f"rank{r}.npz")) for r in (0, 1)]
    assert outs[0]["params_checksum"] == outs[1]["params_checksum"] and np.array_equal(outs[0]["final_params_sub"], outs[1]["final_params_sub"]), \
        "the replicas diverged"
    problems, report = [], []
    for r in (0, 1):
        o = {k: (v.item() if v.ndim == 0 else v) for k, v in outs[r].items()}
        rep = []
        problems += [f"rank {r}: {p}" for p in check_atari_iteration(o, g, {}, sfx=f"_rank{r}", report=rep)]
        report += [f"rank {r}: {x}" for x in rep]
    print("\n".join(["config D whole iteration under update graphs vs the reference's lines:"] + report))
    assert not problems, "\n".join(problems + report)


@pytest.mark.parametrize("N,T,nmb,epochs", [(32, 8, 2, 2), (256, 128, 4, 4)])
def test_update_graphs_with_two_ranks_are_bit_identical_to_the_eager_two_rank_update(tmp_path, N, T, nmb, epochs):
    """``capture_update`` with world = 2: every (epoch, minibatch) slot is THREE hipGraphs -- [forward + loss + backward down to the FC
    weight's gradient] | early bucket | [conv backward] | the rest | [clip + Adam] -- with the three all-reduces of the eager path issued
    between the replays (ppo_atari_multigpu.py:358-377).  Two ranks on one GPU over gloo, each with an eager twin from the same seeds:
    parameters, Adam state and logged scalars bit-equal after every one of three iterations, replicas bit-equal across the ranks -- at a
    small shape and at BASELINE config D's per-GPU size (256 envs x 128 steps, 16 slots of 8,192 rows)."""
    _torchrun([os.path.join("tests", "dp_graphs_worker.py"), str(tmp_path), str(N), str(T), str(nmb), str(epochs), "3"], timeout=1200)
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in (0, 1))
    for r in (r0, r1):
        assert list(r["segs"]) == [3] and bool(r["early"]), "a slot was not cut at both bucket boundaries"
        assert bool(r["self_check"]) and bool(r["self_check_restored"]), "the start-up self-check (captured vs eager update, state restored)"
        assert r["same"].all(), f"captured and eager learners diverged: {r['same']}"
        assert r["scalars_same"].all()
        assert np.array_equal(r["params"], r["params_eager"])
    assert np.array_equal(r0["params"], r1["params"]), "the replicas diverged"


def test_update_graphs_in_the_reference_arrangement_two_graphs_per_slot_self_checked(tmp_path):
    """What ``learner.update_graph_policy`` / ``early_bucket_policy`` choose over RCCL by default (round 6), run here over gloo: NO early bucket -- the
    reference's single all-reduce of the flat gradient behind the backward (ppo_atari_multigpu.py:360-367) --, so a slot is TWO graphs cut on the calling
    thread; before use, ``self_check_update_graphs`` (one captured against one eager update from the same state and permutations, collectives included)
    must report bit-identity and leave parameters, Adam state, gradients, step counter and numpy's stream as they were.  Then the usual: two ranks, each
    with an eager twin, bit-equal after every iteration, replicas equal."""
    _torchrun([os.path.join("tests", "dp_graphs_worker.py"), str(tmp_path), "32", "8", "2", "2", "2", "noearly"], timeout=1200)
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in (0, 1))
    for r in (r0, r1):
        assert list(r["segs"]) == [2] and not bool(r["early"]), "the slots were not captured in the two-graph form"
        assert bool(r["self_check"]) and bool(r["self_check_restored"])
        assert r["same"].all() and r["scalars_same"].all() and np.array_equal(r["params"], r["params_eager"])
    assert np.array_equal(r0["params"], r1["params"]), "the replicas diverged"


def test_update_graphs_with_four_ranks_are_bit_identical_to_the_eager_four_rank_update(tmp_path):
    """The same at world = 4 (round 6: nothing may assume two ranks -- ``grad_scale`` = 1/4, seeds ``seed + rank``, the three all-reduce pieces and
    their offsets, the MIN agreement): four ranks on one GPU over gloo, each with an eager twin; every rank's captured learner bit-equal
    to its twin after each of two iterations, the four replicas bit-equal to one another."""
    _torchrun([os.path.join("tests", "dp_graphs_worker.py"), str(tmp_path), "16", "8", "2", "2", "2"], timeout=1200, nproc=4)
    rs = [np.load(tmp_path / f"rank{r}.npz") for r in range(4)]
    for r in rs:
        assert list(r["segs"]) == [3] and bool(r["early"]) and r["same"].all() and r["scalars_same"].all()
        assert np.array_equal(r["params"], r["params_eager"])
        assert np.array_equal(r["params"], rs[0]["params"]), "the replicas diverged"


@pytest.mark.parametrize("world", [2, 4])
def test_peer_memory_all_reduce_is_the_rank_order_sum_on_every_rank(tmp_path, world):
    """The exchange that needs no collective library (round 6; include/mi355ppo.h a9/e, csrc/dpcomm.hip -- SURVEY 8b's second cut of the all-reduce
    block ppo_atari_multigpu.py:360-367): 2 / 4 PROCESSES on one GPU, every rank's segment mapped by its peers through HIP IPC.  Checked bit for
    bit against ((g0 + g1) + g2) + g3 in f32: 14 sizes from 1 float to the agent's 1,686,693 (partial vectors, slices shorter than a vector, guard
    words untouched), 300 rounds back to back without a host synchronisation and with one rank held up at a time (the one-buffer protocol), the
    call captured in a hipGraph and replayed 12 times with eager calls in between (the round counter lives on the device)."""
    import json

    _torchrun([os.path.join("tests", "dp_peer_worker.py"), str(tmp_path), "raw"], nproc=world, timeout=900)
    rs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    for r, j in enumerate(rs):
        bad = [n for n, ok in j["sizes"].items() if not ok]
        assert not bad, f"rank {r}: sizes {bad} differ from the rank-order sum"
        assert j["back_to_back"], f"rank {r}: a round of the back-to-back run differs"
        assert j["graph_replays"], f"rank {r}: a replayed exchange differs"
        assert j["status"] is None, f"rank {r}: a wait gave up: {j['status']}"
    print(f"world {world}: {max(j['us_per_exchange_same_device'] for j in rs):.0f} us per exchange of 6.75 MB, all ranks on one device (no fabric)")


def test_peer_memory_all_reduce_gives_up_on_a_missing_rank_instead_of_hanging(tmp_path):
    """Rank 1 skips a call: rank 0's wait ends after the communicator's timeout (1.5 s here), the round / peer / phase are recorded in host-visible
    memory, later calls on the broken communicator return at once and ``check()`` raises -- the device is never left spinning."""
    import json

    _torchrun([os.path.join("tests", "dp_peer_worker.py"), str(tmp_path), "timeout"], nproc=2, timeout=300)
    r0, r1 = (json.load(open(tmp_path / f"rank{r}.json")) for r in (0, 1))
    assert r0["first_ok"] and r1["first_ok"]
    assert r0["status"] is not None and r0["status"][0] == 2 and r0["status"][1] == 1 and r0["status"][2] == "reduce" and r0["raised"], r0
    assert 1.0 < r0["waited_s"] < 10.0, r0
    assert r1["status"] is None and not r1["raised"]


def test_update_over_peer_memory_one_graph_per_slot_bit_identical_to_the_process_group_route(tmp_path):
    """``MI355PPO_ALLREDUCE=peer`` in ``PPOLearner`` with world = 2: the flat gradient crosses HIP IPC segments by five small launches on the compute
    stream, so a captured slot is ONE hipGraph with the exchange inside (the process group's all-reduce of the gradient is never called: the worker
    makes it raise).  Each rank's captured learner against its eager twin after every iteration, the replicas against each other, the start-up
    self-check -- and, because g0 + g1 is the same f32 sum whoever computes it, bit-identical parameters to the two-graph route over gloo from the
    same seeds (ppo_atari_multigpu.py:358-377)."""
    (tmp_path / "peer").mkdir(); (tmp_path / "pg").mkdir()
    _torchrun([os.path.join("tests", "dp_graphs_worker.py"), str(tmp_path / "peer"), "32", "8", "2", "2", "2", "peer"], timeout=1200)
    _torchrun([os.path.join("tests", "dp_graphs_worker.py"), str(tmp_path / "pg"), "32", "8", "2", "2", "2", "noearly"], timeout=1200)
    r0, r1 = (np.load(tmp_path / "peer" / f"rank{r}.npz") for r in (0, 1))
    for r in (r0, r1):
        assert list(r["segs"]) == [1] and not bool(r["early"]), "a slot over peer memory is one graph"
        assert bool(r["self_check"]) and bool(r["self_check_restored"]) and bool(r["peer_status_ok"])
        assert r["same"].all() and r["scalars_same"].all() and np.array_equal(r["params"], r["params_eager"])
        assert float(r["moved"]) > 0.9
    assert np.array_equal(r0["params"], r1["params"]), "the replicas diverged"
    assert np.array_equal(r0["params"], np.load(tmp_path / "pg" / "rank0.npz")["params"]), "peer-memory and process-group routes differ at world = 2"


def test_update_over_peer_memory_with_four_ranks(tmp_path):
    """The same at world = 4 (slices of a quarter, ((g0 + g1) + g2) + g3 on the owning rank): captured = eager on every rank, replicas bit-equal."""
    _torchrun([os.path.join("tests", "dp_graphs_worker.py"), str(tmp_path), "16", "8", "2", "2", "2", "peer"], timeout=1200, nproc=4)
    rs = [np.load(tmp_path / f"rank{r}.npz") for r in range(4)]
    for r in rs:
        assert list(r["segs"]) == [1] and r["same"].all() and r["scalars_same"].all() and bool(r["peer_status_ok"])
        assert np.array_equal(r["params"], r["params_eager"]) and np.array_equal(r["params"], rs[0]["params"]), "the replicas diverged"


def test_ppo_atari_multigpu_script_four_ranks_on_one_gpu():
    """The drop-in script with FOUR ranks on device 0 over gloo (ppo_atari_multigpu.py:166-177,360-377 at world = 4): replicas print the same
    actor weight sum after every update; under the default policy (gloo: graphs) every rank replays the cut graphs."""
    out = _torchrun([os.path.join("cleanrl_amd", "ppo_atari_multigpu.py"), "--cuda", "--backend", "gloo", "--device-ids", "0", "0", "0", "0",
                     "--local-num-envs", "4", "--num-steps", "8", "--num-envs", "16", "--total-timesteps", "256"], nproc=4)
    pat = r"local_rank: (\d+), action\.sum\(\): (-?\d+), iteration: (\d+), agent\.actor\.weight\.sum\(\): (-?[\d.eE+-]+)"
    sums = {}
    for lr, a, it, w in re.findall(pat, out):
        sums.setdefault(it, {})[lr] = w
    assert len(sums) == 2, out[-2000:]
    for it, by_rank in sums.items():
        assert len(by_rank) == 4 and len(set(by_rank.values())) == 1, f"replicas diverged at iteration {it}: {by_rank}"
    assert "eager launches" not in out, out[-2000:]


def test_ppo_atari_multigpu_script_over_peer_memory_two_ranks_on_one_gpu():
    """The drop-in script with ``MI355PPO_ALLREDUCE=peer``: nothing else changes on the command line (ppo_atari_multigpu.py:166-177 stays the
    rendezvous); the replicas print the same actor weight sum after every update -- and the same sums as the process-group route (world = 2: the
    same f32 additions)."""
    args = [os.path.join("cleanrl_amd", "ppo_atari_multigpu.py"), "--cuda", "--backend", "gloo", "--device-ids", "0", "0",
            "--local-num-envs", "4", "--num-steps", "8", "--num-envs", "8", "--total-timesteps", "192"]
    pat = r"local_rank: (\d+), action\.sum\(\): (-?\d+), iteration: (\d+), agent\.actor\.weight\.sum\(\): (-?[\d.eE+-]+)"
    by_route = {}
    for route in ("peer", "pg"):
        out = _torchrun(args, extra_env={"MI355PPO_ALLREDUCE": route})
        sums = {}
        for lr, a, it, w in re.findall(pat, out):
            sums.setdefault(it, {})[lr] = w
        assert len(sums) == 3, out[-2000:]
        for it, by_rank in sums.items():
            assert len(by_rank) == 2 and by_rank["0"] == by_rank["1"], f"{route}: replicas diverged at iteration {it}: {by_rank}"
        assert "eager launches" not in out, out[-2000:]
        by_route[route] = {it: v["0"] for it, v in sums.items()}
    assert by_route["peer"] == by_route["pg"], by_route


def test_ppo_atari_multigpu_script_two_ranks_on_one_gpu():
    """The drop-in script itself, ``--cuda`` on, both ranks on device 0 (``--device-ids 0 0``), backend gloo: replicas print
    the same actor weight sum after every update while sampling different actions (per-rank seeds)."""
    out = _torchrun([os.path.join("cleanrl_amd", "ppo_atari_multigpu.py"), "--cuda", "--backend", "gloo", "--device-ids", "0", "0",
                     "--local-num-envs", "4", "--num-steps", "8", "--num-envs", "8", "--total-timesteps", "192"])
    pat = r"local_rank: (\d+), action\.sum\(\): (-?\d+), iteration: (\d+), agent\.actor\.weight\.sum\(\): (-?[\d.eE+-]+)"
    sums, acts = {}, {}
    for lr, a, it, w in re.findall(pat, out):
        sums.setdefault(it, {})[lr] = w
        acts.setdefault(it, {})[lr] = a
    assert len(sums) == 3, out[-2000:]
    for it, by_rank in sums.items():
        assert len(by_rank) == 2 and by_rank["0"] == by_rank["1"], f"replicas diverged at iteration {it}: {by_rank}"
    assert len({sums[i]["0"] for i in sums}) == 3                     # the weights move every iteration
    assert any(acts[i]["0"] != acts[i]["1"] for i in acts)           # different rollouts per rank


@pytest.mark.parametrize("world", [2, 4])
def test_bench_multi_rank_legs_run_with_several_ranks_on_one_gpu(world):
    """``bench.py --gpus 2 / 4`` end to end with both ranks on cuda:0 (``--same-device --backend gloo``: a plumbing smoke, not a
    measurement): the self-launcher, the rendezvous on 127.0.0.1, the barrier + synchronize brackets, the max-over-ranks
    all-reduce of the elapsed time and the rank-0 JSON line -- everything the driver's 2/4/8-GPU runs go through except RCCL
    itself (reference launch shape: ppo_atari_multigpu.py:166-177; its all-reduce :360-377)."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "1", "--same-device",
           "--backend", "gloo", "--local-num-envs", "64", "--num-steps", "16", "--no-cpu-baseline", "--no-pcie-inclusive"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + "\n" + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line (rank 0 only), got {len(lines)}:\n{out.stdout[-2000:]}"
    j = json.loads(lines[0])
    assert j["n_gpus"] == world and j["steps"] == 1 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["config"]["parallelism"].startswith(f"dp{world}") and "SAME-DEVICE" in j["config"]["parallelism"]
    assert j["config"]["global_num_envs"] == 64 * world and j["config"]["local_num_envs"] == 64
    assert np.isfinite(j["final_loss"]) and j["value"] > 0
    # the first collective of the run was checked element by element (round 6: bench.py::preflight)
    assert j["collective_preflight"]["world"] == world and j["collective_preflight"]["bytes"] == 4 * 1686693
    assert "three hipGraphs" in j["config"]["update"]                    # gloo: the cut graphs, agreed on by all ranks
    assert "preflight ok" in out.stderr
    # value = the units ALL ranks processed / the max-over-ranks time
    np.testing.assert_allclose(j["value"], world * 64 * 16 * 1 / (j["ms_per_step"] * 1e-3), rtol=1e-6)
    if world == 2:
        # --preflight: only the process-group check, one JSON line, exit 0
        r = subprocess.run(cmd[:2] + ["--gpus", "2", "--same-device", "--backend", "gloo", "--preflight"], cwd=ROOT, capture_output=True, text=True,
                           timeout=300, env=env)
        assert r.returncode == 0 and '"preflight"' in r.stdout and "preflight ok" in r.stderr, r.stdout[-2000:] + r.stderr[-2000:]
    # without --same-device the launcher refuses to run fewer ranks than asked on a one-GPU box
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT,
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 2 and "refusing" in r.stderr


def test_bench_two_ranks_over_peer_memory_on_one_gpu():
    """``MI355PPO_ALLREDUCE=peer python bench.py --gpus 2 --same-device --backend gloo``: the line says which route ran -- one hipGraph per slot with
    the exchange inside, gradients over HIP IPC segments -- and the run ends with a finite loss and no timeout."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MI355PPO_ALLREDUCE"] = "peer"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--same-device",
           "--backend", "gloo", "--local-num-envs", "64", "--num-steps", "16", "--no-cpu-baseline", "--no-pcie-inclusive"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + "\n" + out.stderr[-3000:]
    j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert "gradient exchange INSIDE" in j["config"]["update"] and "HIP IPC" in j["config"]["parallelism"], j["config"]
    assert np.isfinite(j["final_loss"]) and j["value"] > 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the first multi-GPU box that runs this suite exercises RCCL")
def test_rccl_two_gpus_readiness_guard():
    """The first time this suite runs on a box with >= 2 GPUs: ``bench.py --gpus 2`` over RCCL (backend nccl), one rank per GPU --
    ``init_process_group("nccl", device_id=...)``, the three-piece all-reduce with the early FC-weight bucket overlapped with the
    conv backward, rollout-graph capture under the process group's watchdog thread, the barrier + max-over-ranks timing -- before
    any scaling number is attempted (reference: ppo_atari_multigpu.py:174-175,360-374).  Nothing N > 1 has run over RCCL on the
    one-GPU boxes of the build rounds; this is the guard the round-3 review asked for."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "nccl", "--steps", "1", "--warmup", "1",
           "--local-num-envs", "64", "--num-steps", "16", "--no-cpu-baseline", "--no-pcie-inclusive"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + "\n" + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and "RCCL" in j["config"]["parallelism"] and np.isfinite(j["final_loss"]) and j["value"] > 0
    upd = j["config"]["update"]
    print("update over RCCL (default policy):", upd)
    # the default over RCCL: two graphs per slot behind the self-check, or -- if the capture or the check failed on a rank -- everybody eager: both are a pass
    assert j["collective_preflight"]["backend"] == "nccl" and ("two hipGraphs" in upd or "self-check failed on a rank" in upd), upd
    # opting in to the early bucket (three graphs, the cut in the middle of the backward) goes through the same self-check and agreement
    env["MI355PPO_UPDATE_GRAPHS"] = "1"
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + "\n" + out.stderr[-3000:]
    env["MI355PPO_UPDATE_GRAPHS"] = "0"
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "eager launches (MI355PPO_UPDATE_GRAPHS=0)" in out.stdout, out.stdout[-3000:] + "\n" + out.stderr[-3000:]
