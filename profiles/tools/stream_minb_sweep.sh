cd $GRAFT_REPO_ROOT
for cfg in "56 4 2" "128 4 3" "48 2 2"; do
  set -- $cfg
  for mb in 48 80 128 1000; do
    line=$(SIND_SOR_STREAM_MINB=$mb timeout -k 10 300 python3 bench.py --streams $1 --frames-per-step $2 --steps 12 --warmup 3 --flow-slices $3 --no-cpu-baseline --no-sequence-leg 2>/dev/null | tail -1)
    echo "$line" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step']
print('%3d pairs/step (%2d streams x %d), %d slices, stream kernel from %4d images: %7.1f pairs/s  %6.1f ms/step  dense flow %6.1f  tails %6.1f' % (d['config']['frame_pairs_per_step'], $1, $2, $3, $mb, d['value'], d['ms_per_step'], st['dense_flow'], st['tails']))"
  done
done
