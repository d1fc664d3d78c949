#!/bin/bash
O=gpurun_out/r4e20; mkdir -p $O
V=tools/scratch/variants
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE"
for t in new wm16off; do
  if [ $t = new ]; then unset PNNP_LIB; else export PNNP_LIB=/root/repo/$V/libpnnp_$t.so; fi
  bash tools/pmc_layers.sh r4e20_$t "$C" --x3 --only wgrad --reps 2 --layers conv2_2,conv4_2 > /dev/null 2>&1
  cp gpurun_out/pmc_layers_r4e20_$t.csv $O/
done
echo done > $O/done.txt
