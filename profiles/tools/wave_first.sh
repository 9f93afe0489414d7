#!/bin/bash
# first measurement of k_sor_wave: parity tests, then the flow pyramid alone with the streaming kernel and with one-wave pipelines
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_flow_gpu.py -x -q -k "wave or streaming_solver_equals" > gpurun_out/wave_tests.txt 2>&1 || { tail -30 gpurun_out/wave_tests.txt; exit 1; }
tail -3 gpurun_out/wave_tests.txt
{
WAVE=0 timeout -k 10 120 python3 profiles/tools/flow_slices_alone.py 3 170 3
for it in 512 1024 2048; do WAVE=1 WAVE_ITEMS=$it timeout -k 10 120 python3 profiles/tools/flow_slices_alone.py 3 170 3; done
WAVE=0 timeout -k 10 120 python3 profiles/tools/flow_slices_alone.py 1 512 3
for it in 1024 2048 4096; do WAVE=1 WAVE_ITEMS=$it timeout -k 10 120 python3 profiles/tools/flow_slices_alone.py 1 512 3; done
} 2>&1 | tee gpurun_out/wave_first.txt
