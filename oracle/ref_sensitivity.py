"""How far do two runs of the REFERENCE's own lines drift apart over the 16 updates of the config-B iteration when only the f32
summation order changes?  TEST INFRASTRUCTURE (build container only; needs /root/reference).

    python -m oracle.ref_sensitivity [threads]            # config B (golden minted with 1 thread)
    python -m oracle.ref_sensitivity C [threads]          # config C at its full size (golden minted with 8 threads; default here: 4)
    python -m oracle.ref_sensitivity D [threads]          # config D, two ranks (golden: 4 threads per rank; default here: 2)

Re-executes oracle/mint_goldens.py::mint_atari_iteration_config_b (ppo_atari_envpool.py:217-322, verbatim) with `threads` torch
CPU threads (default 8; the committed golden was minted with 1: oneDNN / ATen then reduce in another order) and prints, for the
quantities tests/test_gpu_learner.py::test_config_b_whole_iteration_teacher_forced_all_16_updates bounds, the distance between
the two reference runs.  The GPU test's bars at updates 8 / 16 are set from these numbers: a kernel path cannot be asked to
follow the reference more closely than the reference follows itself."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from oracle import mint_goldens as MG

    if len(sys.argv) > 1 and sys.argv[1].upper() == "C":
        from oracle import mint_full_size as MF

        threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
        g = dict(np.load(os.path.join(MG.OUT, "atari_iteration_cfgC.npz")))
        g = {k.split("/", 1)[1]: v for k, v in g.items()}
        torch.use_deterministic_algorithms(False)
        d = MF.mint_config_c(threads=threads, save=False)
        out = {"config": "C", "threads": threads, "golden_threads": int(g["torch_threads"])}
    elif len(sys.argv) > 1 and sys.argv[1].upper() == "D":
        from oracle import mint_full_size as MF

        threads = int(sys.argv[2]) if len(sys.argv) > 2 else 2
        g = dict(np.load(os.path.join(MG.OUT, "atari_iteration_cfgD.npz")))
        g = {k.split("/", 1)[1]: v for k, v in g.items()}
        g["values"] = g["values_rank0"]
        torch.use_deterministic_algorithms(False)
        d = MF.mint_config_d(threads=threads, save=False)
        d["values"] = d["values_rank0"]
        out = {"config": "D", "threads": threads, "golden_threads": int(g["torch_threads"])}
    else:
        threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
        g = dict(np.load(os.path.join(MG.OUT, "atari_iteration_cfgB.npz")))
        g = {k.split("/", 1)[1]: v for k, v in g.items()}
        torch.use_deterministic_algorithms(False)
        torch.set_num_threads(threads)
        d = MG.mint_atari_iteration_config_b(save=False)
        out = {"threads": threads}
    cos = lambda a, b: float(np.dot(a.astype(np.float64), b.astype(np.float64)) / (np.linalg.norm(a.astype(np.float64)) * np.linalg.norm(b.astype(np.float64))))
    out["values_max_abs"] = float(np.abs(d["values"] - g["values"]).max())
    sc_err = np.abs(d["scalars"].astype(np.float64) - g["scalars"].astype(np.float64))
    out["scalars_max_abs_by_column"] = sc_err.max(0).tolist()
    out["scalars_max_rel_loss"] = float((sc_err[:, 0] / np.abs(g["scalars"][:, 0])).max())
    for k in (1, 8, 16):
        a, b = d[f"mb{k}_grad_sub"], g[f"mb{k}_grad_sub"]
        out[f"update{k}"] = {"max_abs_over_absmax": float(np.abs(a - b).max() / g[f"mb{k}_grad_absmax"]), "cosine": cos(a, b),
                             "norm_rel": float(d[f"mb{k}_grad_norm"] / g[f"mb{k}_grad_norm"] - 1.0),
                             "per_tensor_norm_rel_max": float(np.abs(d[f"mb{k}_grad_tensor_norms"] / g[f"mb{k}_grad_tensor_norms"] - 1.0).max())}
    da, db = d["final_params_sub"] - d["init_params_sub"], g["final_params_sub"] - g["init_params_sub"]
    out["param_move"] = {"cosine": cos(da, db), "length_ratio": float(np.linalg.norm(da) / np.linalg.norm(db)),
                         "frac_within_5pct": float(np.isclose(da, db, rtol=5e-2, atol=2e-5).mean())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
