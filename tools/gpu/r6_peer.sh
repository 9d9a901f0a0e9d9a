#!/bin/bash
# the peer-memory gradient exchange (csrc/dpcomm.hip) on one GPU with 2 / 4 processes
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r6peer}; mkdir -p $O
timeout ${LIMIT:-1500} python -m pytest tests/test_gpu_multirank.py -q -x -p no:cacheprovider -s -k "${K:-peer}" 2>&1 | tail -${TAIL:-40} | cut -c1-400 | tee $O/pytest_peer.txt
