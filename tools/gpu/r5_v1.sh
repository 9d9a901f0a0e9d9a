#!/bin/bash
# Round 5, visit 1: the two-term f16 split (csrc/f16split.h) for the first time on the GPU -- the f16 MFMA's subnormal check, the per-kernel
# A/B of one minibatch's launches at 32,768 images (bf16x3 vs f16x2, torch-free driver), the f16x2 parity tests, then the CNN / learner
# suites (whose default path is now f16x2) and a bench A/B.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5v1
rm -rf $O; mkdir -p $O
cd $R
T0=$SECONDS
tools/mfma_denorm > $O/mfma_denorm.json 2>&1; cat $O/mfma_denorm.json
for i in 1 2; do
  timeout 120 tools/conv_traffic 32768 4 2>&1 | grep '^{' | sed 's/^/{"split": "bf16x3", /; s/{"split": "bf16x3", {/{"split": "bf16x3", /' >> $O/conv_traffic_ab.jsonl
  CONV_TRAFFIC_F16=1 timeout 120 tools/conv_traffic 32768 4 2>&1 | grep '^{' | sed 's/^{/{"split": "f16x2", /' >> $O/conv_traffic_ab.jsonl
done
cat $O/conv_traffic_ab.jsonl | cut -c1-420
echo "traffic t=$((SECONDS-T0))"
CONV_TRAFFIC_F16=1 timeout 60 tools/conv_traffic 1024 4 2>&1 | grep '^{' | cut -c1-420
timeout 60 tools/conv_traffic 1024 4 2>&1 | grep '^{' | cut -c1-420
(timeout 900 python -m pytest tests/test_gpu_f16x2.py -q -p no:cacheprovider -x --deselect tests/test_gpu_f16x2.py::test_trunk_autograd_under_f16x2_against_float64) > $O/pytest_f16x2.log 2>&1; echo "pytest f16x2 rc=$? t=$((SECONDS-T0))"; tail -25 $O/pytest_f16x2.log | cut -c1-300
(timeout 300 python -m pytest tests/test_gpu_f16x2.py -q -p no:cacheprovider -k trunk_autograd) > $O/pytest_f16x2_trunk.log 2>&1; echo "pytest trunk rc=$? t=$((SECONDS-T0))"; tail -15 $O/pytest_f16x2_trunk.log | cut -c1-300
(timeout 900 python -m pytest tests/test_gpu_cnn.py tests/test_gpu_kernels.py -q -p no:cacheprovider) > $O/pytest_cnn.log 2>&1; echo "pytest cnn rc=$? t=$((SECONDS-T0))"; tail -8 $O/pytest_cnn.log | cut -c1-300
(timeout 1200 python -m pytest tests/test_gpu_learner.py -q -p no:cacheprovider -s) > $O/pytest_learner.log 2>&1; echo "pytest learner rc=$? t=$((SECONDS-T0))"; grep -E "passed|failed|config C|update [0-9]+:|values |capture" $O/pytest_learner.log | tail -30 | cut -c1-400
for sp in bf16x3 f16x2 bf16x3 f16x2; do
  MI355PPO_SPLIT=$sp timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-pcie-inclusive 2> $O/bench_$sp.err | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline', {})
    print(json.dumps({'split': '$sp', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'phases': d.get('phases_ms', d.get('phase_ms')), 'roofline_kernel': r.get('kernel', '')[:60], 'frac': r.get('frac'), 'avg_us': r.get('avg_launch_us'),
                      'kernels': {k: round(v['avg_us'], 1) for k, v in d.get('kernels', {}).items() if isinstance(v, dict) and 'avg_us' in v and k.endswith('32768')}}))
" >> $O/bench_ab.jsonl
  echo "bench $sp t=$((SECONDS-T0))"
done
cat $O/bench_ab.jsonl | cut -c1-900
echo "total t=$((SECONDS-T0))"
