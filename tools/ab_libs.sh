#!/bin/bash
# usage: tools/ab_libs.sh <tag> <rounds> <lib>... : alternate builds of the library (SEQREC_LIB) on the fresh-batch bench; prints the step and the scan calls
out=gpurun_out/$1; rounds=$2; shift 2; mkdir -p $out
for i in $(seq 1 $rounds); do for l in "$@"; do
  b=$(basename $l)
  SEQREC_LIB=$PWD/$l timeout -k 10 300 python bench.py --gpus 1 --steps 200 --warmup 20 --cpu-seconds 0 --recall-steps 0 $AB_ARGS > $out/ab_${b}_$i.log 2> $out/ab_${b}_$i.err || { echo "$b failed"; tail -3 $out/ab_${b}_$i.err; continue; }
  python -c "
import json
d=json.loads(open('$out/ab_${b}_$i.log').read().strip().splitlines()[-1]); print('%-28s' % '$b', $i, d['value'], d['ms_per_step'], d.get('ms_per_step_median'), [(k[11:],x['avg_us']) for k,x in d['kernels'].items() if 'rnn_' in k])"
done; done
