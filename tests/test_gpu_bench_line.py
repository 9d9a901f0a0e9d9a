"""The JSON line of ``bench.py`` on a real GPU at a small configuration (round-5 review, item 4): the fields the contract names, the arithmetic
described from the kernels that ran, the HBM-regime K1 / K3 points with their fractions, the CPU baseline marked with whether it was taken at the
metric's configuration.  (Reference workload: cleanrl/ppo_atari_envpool.py at BASELINE configs[1]'s shape.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_says_what_ran_at_the_size_it_claims():
    env = {k: v for k, v in os.environ.items() if not k.startswith("MI355PPO_")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "B", "--steps", "1", "--warmup", "1", "--no-pcie-inclusive", "--cpu-baseline-envs", "4",
           "--cpu-baseline-full", "off"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + "\n" + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "matrix_arithmetic"):
        assert k in j, k
    assert j["dtype"] == "f32" and "two-term f16 split" in j["matrix_arithmetic"] and "f32 accumulate" in j["matrix_arithmetic"]
    cnn = j["config"]["cnn"]
    assert "kernel Q" in cnn and "bf16 MFMA over exact three-term" not in cnn
    for launch in ("conv2_fwd R", "conv2_dgrad R", "conv1_wgrad U", "fc_dgrad G"):          # letters of the kernels that ran at 4,096 rows
        assert launch in cnn, (launch, cnn)
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    g, l = j["kernels"]["gae_hbm_regime"], j["kernels"]["loss_hbm_regime"]
    assert g["N"] == 1 << 20 and g["algorithmic_bytes"] == 20 * 128 * (1 << 20) + 8 * (1 << 20) and 0.3 < g["frac"] < 1.0
    assert l["M"] == 1 << 22 and 0.2 < l["streaming"]["frac"] < 1.0 and 0.05 < l["permuted"]["frac"] < 1.0 and l["frac"] == l["streaming"]["frac"]
    p = j["hbm_stream_probe"]          # what plain streams reach on the box (tools/hbm_probe quick), beside the 8 TB/s the fractions are quoted against
    assert p["peak_GBps"] == 8000.0 and all(1000.0 < p[k] < 8000.0 for k in ("read_only_GBps", "write_only_GBps", "copy_GBps", "mix_1r_2w_GBps"))
    cb = j["cpu_baseline"]
    # (config B, --cpu-baseline-full off: the small whole-iteration sample; the default line -- config C -- takes the bounded sample at the metric's shapes)
    assert cb["kind"] == "port" and cb["at_metric_config"] is False and cb["value"] > 0 and "cross_check_8_cores_port_value" in cb
