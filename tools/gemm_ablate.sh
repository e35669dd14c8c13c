#!/bin/bash
# timing-only ablations of the LDS-DMA GEMM (diagnostic build tools/bin/libseqrec_ablate.so, -DSEQREC_GEMM_ABLATE):
# mask bits 1 no MFMA, 2 no in-loop DMA, 4 no C stores, 8 no LDS fragment reads
out=gpurun_out/$1; mkdir -p $out
for m in 0 1 2 4 8 3 6 12 14 15; do
  echo "== ablate mask $m" >> $out/ablate.log
  SEQREC_LIB=$PWD/tools/bin/libseqrec_ablate.so SEQREC_GEMM_ABLATE=$m ONLY=${ONLY:-logits,dEneg,sat-logits} TILES=${TILES:-1,2} timeout -k 10 120 python tools/bench_gemm2.py >> $out/ablate.log 2>&1 || exit 1
done
grep -v amdgpu.ids $out/ablate.log
