cd $GRAFT_REPO_ROOT
meas() { cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_ab_$1 -o b1 -- python3 $GRAFT_REPO_ROOT/profiles/tools/dropin_latency.py --flow-only --batches 1 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; echo "== $1"; python3 profiles/tools/db_kernel_stats.py gpurun_out/prof_ab_$1/b1_results.db 13 3 | tail -3; rm -rf gpurun_out/prof_ab_$1; }
meas packed_1
cd sindslam_amd/csrc && hipcc -O3 -march=x86-64-v3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-result -I../../include -fno-slp-vectorize -c flow_coarse.hip -o build/flow_coarse.o && hipcc -shared -o ../libsind_hip.so build/*.o build/host/*.o --offload-arch=gfx950 -ldl && cd ../..
meas unpacked_1
timeout 300 python -m pytest tests/test_flow_coarse_gpu.py -x -q 2>&1 | tail -1
cd sindslam_amd/csrc && touch flow_coarse.hip && make -s all && cd ../..
meas packed_2
