#!/bin/bash
# does host load slow the flow's launch threads?  three slices of 170 pairs alone, then beside N busy-looping processes (the box's quota is 16 cores)
mkdir -p gpurun_out
python3 profiles/tools/flow_slices_alone.py 3 170 4 2>/dev/null | tail -1
for n in 8 12 15; do
  pids=""
  for i in $(seq $n); do python3 -c "
import time
t=time.time()
while time.time()-t<60: pass" & pids="$pids $!"; done
  sleep 1
  echo "with $n busy processes: $(python3 profiles/tools/flow_slices_alone.py 3 170 4 2>/dev/null | tail -1)"
  kill $pids 2>/dev/null; wait 2>/dev/null
done
