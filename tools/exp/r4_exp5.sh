#!/bin/bash
# round 4, experiment 5: epilogue phases (stamps), fair SKIP_STORE bound, stores dropped
O=gpurun_out/r4e5; mkdir -p $O
V=tools/scratch/variants
for shp in "512 32 32" "512 64 32" "256 64 64" "64 256 256"; do
  echo "== stamps $shp" >> $O/stamps.txt
  PNNP_LIB=$V/libpnnp_stampsn.so python tools/x3_stamps.py $shp 2>&1 | grep -v "^/opt" >> $O/stamps.txt
done
for r in 1 2; do
  for t in new skipst skipst_nobar epioob; do
    echo "== $t" >> $O/layers.txt
    if [ $t = new ]; then unset PNNP_LIB; else export PNNP_LIB=$V/libpnnp_$t.so; fi
    python tools/layer_bench.py --x3 --only fwd,dgrad --reps 7 2>/dev/null | grep -v "^layer" >> $O/layers.txt
  done
done
echo done > $O/done.txt
