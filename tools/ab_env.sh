#!/bin/bash
# usage: tools/ab_env.sh <tag> VAR valA valB : alternate two settings of an env switch on the fresh-batch bench, print kernel of interest
out=gpurun_out/$1; mkdir -p $out; var=$2
for i in 1 2; do for v in $3 $4; do
  env $var=$v SEQREC_SCAN_GRAPH=0 timeout -k 10 300 python bench.py --gpus 1 --steps 300 --warmup 20 --cpu-seconds 0 --recall-steps 0 --tune-steps 0 > $out/ab_${var}_${v}_$i.log 2> $out/ab_${var}_${v}_$i.err
  python -c "
import json
d=json.loads(open('$out/ab_${var}_${v}_$i.log').read().strip().splitlines()[-1]); print('$var=$v', $i, d['value'], d['ms_per_step'], [(k[7:],x['avg_us']) for k,x in d['kernels'].items() if '$5' in k])"
done; done
