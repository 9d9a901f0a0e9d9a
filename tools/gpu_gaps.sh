export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_gaps -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie-inclusive > $GRAFT_REPO_ROOT/gpurun_out/prof_gaps.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
db=$(find gpurun_out/prof_gaps -name '*.db' | head -1); python tools/rocpd_gaps.py "$db" | tee gpurun_out/gaps.txt
rm -rf gpurun_out/prof_gaps
