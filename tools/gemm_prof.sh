#!/bin/bash
# ground-truth kernel durations (rocprofv3 --kernel-trace --stats) of the ablation masks on one shape
out=$PWD/gpurun_out/$1; mkdir -p $out; root=$PWD
cd /tmp; export TMPDIR=/tmp
for m in ${MASKS:-0 15 14 3 1}; do
  export SEQREC_LIB=$root/tools/bin/libseqrec_ablate.so SEQREC_GEMM_ABLATE=$m ONLY=${ONLY:-logits} TILES=${TILES:-1}
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$m -o p -- python3 $root/tools/bench_gemm2.py > $out/p$m.log 2>&1 || { echo "mask $m failed"; tail -3 $out/p$m.log; exit 1; }
  f=$(find $out/p$m -name '*kernel_stats.csv' | head -1)
  echo "== mask $m"; grep -v amdgpu.ids $out/p$m.log | grep med; grep gemm2 $f | awk -F, '{print "   kernel-trace avg ns:", $4, " calls:", $2, " min:", $6, " max:", $7}'
  rm -rf $out/p$m
done
