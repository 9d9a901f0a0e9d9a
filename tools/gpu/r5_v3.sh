#!/bin/bash
# Round 5, visit 3: after the SGPR wait-state fix -- f16x2 parity tests, the CNN / kernel / learner suites, bench A/B
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5v3
rm -rf $O; mkdir -p $O
cd $R
T0=$SECONDS
(timeout 900 python -m pytest tests/test_gpu_f16x2.py -q -p no:cacheprovider) > $O/pytest_f16x2.log 2>&1; echo "pytest f16x2 rc=$? t=$((SECONDS-T0))"
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_f16x2.log | cut -c1-250
grep -E "^E  +(AssertionError|.*Error)" $O/pytest_f16x2.log | cut -c1-400 | head -30
(timeout 900 python -m pytest tests/test_gpu_cnn.py tests/test_gpu_kernels.py -q -p no:cacheprovider) > $O/pytest_cnn.log 2>&1; echo "pytest cnn rc=$? t=$((SECONDS-T0))"
grep -E "^(FAILED|ERROR)|passed|failed|^E  +Assert" $O/pytest_cnn.log | cut -c1-300
(timeout 1500 python -m pytest tests/test_gpu_learner.py -q -p no:cacheprovider -s) > $O/pytest_learner.log 2>&1; echo "pytest learner rc=$? t=$((SECONDS-T0))"
grep -E "^(FAILED|ERROR)|passed|failed|config C|update [0-9]+:|values |capture with|^E  +[A-Za-z]*Error" $O/pytest_learner.log | tail -40 | cut -c1-400
for sp in bf16x3 f16x2; do
  MI355PPO_SPLIT=$sp timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-pcie-inclusive 2> $O/bench_$sp.err | grep '^{' > $O/bench_$sp.json
  python - <<PY >> $O/bench_ab.jsonl
import json
d = json.loads(open("$O/bench_$sp.json").readline()); r = d.get("roofline", {})
print(json.dumps({"split": "$sp", "value": d["value"], "ms_per_step": d["ms_per_step"], "phases": d.get("phases_ms"), "frac": r.get("frac"), "avg_us": r.get("avg_launch_us"),
                  "kernels": {k: round(v["avg_us"], 1) for k, v in d.get("kernels", {}).items() if isinstance(v, dict) and "avg_us" in v and k.endswith("32768")}}))
PY
  echo "bench $sp t=$((SECONDS-T0))"
done
cat $O/bench_ab.jsonl | cut -c1-700
echo "total t=$((SECONDS-T0))"
