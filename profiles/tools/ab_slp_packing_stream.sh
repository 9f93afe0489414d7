# A/B inside one call: flow_kernels.hip (k_sor_stream, k_coef_lanes, k_sor_fused) with and without the SLP vectoriser; the solver alone on 170 pairs
cd $GRAFT_REPO_ROOT
meas() { cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_ab_$1 -o b1 -- python3 $GRAFT_REPO_ROOT/profiles/tools/sor_only.py 170 2 384 288 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; echo "== $1"; python3 profiles/tools/db_kernel_stats.py gpurun_out/prof_ab_$1/b1_results.db 3 4 | tail -4; rm -rf gpurun_out/prof_ab_$1; }
meas packed_1
cd sindslam_amd/csrc && hipcc -O3 -march=x86-64-v3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-result -I../../include -fno-slp-vectorize -c flow_kernels.hip -o build/flow_kernels.o && hipcc -shared -o ../libsind_hip.so build/*.o build/host/*.o --offload-arch=gfx950 -ldl && cd ../..
meas unpacked_1
timeout 600 python -m pytest tests/test_flow_gpu.py -x -q 2>&1 | tail -1
cd sindslam_amd/csrc && touch flow_kernels.hip && make -s all && cd ../..
meas packed_2
