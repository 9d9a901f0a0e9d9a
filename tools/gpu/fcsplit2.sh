#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
for w in 512 1024 1536 2048 3072; do for m in 1024 512; do echo -n "waves=$w "; MI355PPO_FC_SPLIT_WAVES=$w timeout 60 tools/conv_traffic $m 8 2>&1 | head -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['images'], d['fc_fwd_us'])"; done; done
