#!/bin/bash
# Round 5, visit 2: the f16x2 parity tests, all of them (no -x), then the trunk tests of the CNN suite
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5v2
rm -rf $O; mkdir -p $O
cd $R
T0=$SECONDS
(timeout 900 python -m pytest tests/test_gpu_f16x2.py -q -p no:cacheprovider) > $O/pytest_f16x2.log 2>&1; echo "pytest f16x2 rc=$? t=$((SECONDS-T0))"
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_f16x2.log | cut -c1-250
grep -E "^E  +(AssertionError|assert|.*Error)" $O/pytest_f16x2.log | cut -c1-300 | head -60
(timeout 600 python -m pytest tests/test_gpu_cnn.py -q -p no:cacheprovider -k "trunk or 4GiB") > $O/pytest_cnn_trunk.log 2>&1; echo "pytest cnn trunk rc=$? t=$((SECONDS-T0))"
grep -E "^(FAILED|ERROR)|passed|failed|^E  +Assert" $O/pytest_cnn_trunk.log | cut -c1-300
echo "total t=$((SECONDS-T0))"
