#!/bin/bash
# usage: tools/ab_lib.sh <tag> <libA> <libB> : alternate two builds of the library (SEQREC_LIB) on the fresh-batch bench, eager issue
out=gpurun_out/$1; mkdir -p $out
for i in 1 2 3; do for l in $2 $3; do
  SEQREC_LIB=$PWD/$l SEQREC_SCAN_GRAPH=0 timeout -k 10 300 python bench.py --gpus 1 --steps 300 --warmup 20 --cpu-seconds 0 --recall-steps 0 --tune-steps 0 > $out/ab_$(basename $l)_$i.log 2> $out/ab_$(basename $l)_$i.err
  python -c "
import json
d=json.loads(open('$out/ab_$(basename $l)_$i.log').read().strip().splitlines()[-1]); print('$(basename $l)', $i, d['value'], d['ms_per_step'], [(k[11:],x['avg_us']) for k,x in d['kernels'].items() if 'rnn_' in k])"
done; done
