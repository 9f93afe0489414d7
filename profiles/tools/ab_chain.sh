# one-call measurement of the small-batch kernels: flow alone at one pair under the kernel trace (tests first)
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_flow_coarse_gpu.py -x -q 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_ab_x -o b1 -- python3 $GRAFT_REPO_ROOT/profiles/tools/dropin_latency.py --flow-only --batches 1 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python3 profiles/tools/db_kernel_stats.py gpurun_out/prof_ab_x/b1_results.db 13 5 | tail -6; rm -rf gpurun_out/prof_ab_x
