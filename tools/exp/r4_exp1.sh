#!/bin/bash
# round 4, experiment 1 (on the GPU box): where the halo staging sits (X3_FILLMODE), what the barriers / the staging cost at most
O=gpurun_out/r4e1; mkdir -p $O
V=tools/scratch/variants
L="--layers conv2_2,conv3_2,conv4_2,conv5_2,conv7_1"
rocprofv3 -L 2>/dev/null | grep -o "SQ_VALU_MFMA_COEXEC_CYCLES\|SQ_WAIT_ANY\b\|SQ_WAIT_INST_ANY\|SQ_ACTIVE_INST_ANY" | sort | uniq -c > $O/counters_avail.txt
# correctness of the new modes
for t in fm1 fm3; do
  PNNP_LIB=$V/libpnnp_$t.so timeout 600 python -m pytest tests/test_gpu_x3.py -x -q -k "fwd or bwd_data or accurate or maxpool" 2>&1 | tail -3 > $O/pytest_$t.txt
done
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_noiseflow.py -x -q 2>&1 | tail -5 > $O/pytest_default.txt
for r in 1 2; do
  for t in fm0 fm1 fm2 fm3 nobar skipst nobar_skipst; do
    echo "== $t" >> $O/layers.txt
    PNNP_LIB=$V/libpnnp_$t.so python tools/layer_bench.py --x3 --only fwd,dgrad --reps 7 $L 2>/dev/null | grep -v "^layer" >> $O/layers.txt
  done
done
for t in stamps0 stamps1 stamps3; do
  echo "== $t" >> $O/stamps.txt
  PNNP_LIB=$V/libpnnp_$t.so python tools/x3_stamps.py 64 256 256 >> $O/stamps.txt 2>&1
done
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"
for t in fm0 fm1 fm3; do
  export PNNP_LIB=/root/repo/$V/libpnnp_$t.so
  bash tools/pmc_layers.sh r4e1_$t "$C" --x3 --only fwd --reps 2 $L > /dev/null 2>&1
  cp gpurun_out/pmc_layers_r4e1_$t.csv $O/ 2>/dev/null
done
unset PNNP_LIB
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_round tools/ubench/mfma_round.hip 2>/dev/null && /tmp/mfma_round > $O/mfma_round.txt 2>&1
echo done > $O/done.txt
