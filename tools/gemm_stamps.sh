#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
export SEQREC_LIB=$PWD/tools/bin/libseqrec_ablate.so
for g in ${GRIDS:-100000 512 256}; do for m in ${MASKS:-0 14}; do
  SEQREC_GEMM_V2_GRID=$g SEQREC_GEMM_ABLATE=$m timeout -k 10 100 python tools/gemm_stamps.py ${SHAPE:-logits} ${TILE:-1} >> $out/stamps.log 2>&1 || exit 1
done; done
grep -v amdgpu.ids $out/stamps.log
