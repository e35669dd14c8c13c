#!/bin/bash
# usage: tools/ab_env2.sh <tag> <rounds> VAR valA valB [bench args...] : alternate two settings of an env switch on the fresh-batch bench (step, median, parity)
out=gpurun_out/$1; rounds=$2; var=$3; a=$4; b=$5; shift 5; mkdir -p $out
for i in $(seq 1 $rounds); do for v in $a $b; do
  env $var=$v timeout -k 10 300 python bench.py --gpus 1 --steps 300 --warmup 20 --cpu-seconds 0 --recall-steps 0 "$@" > $out/ab_${var}_${v}_$i.log 2> $out/ab_${var}_${v}_$i.err || { echo "$var=$v failed"; tail -3 $out/ab_${var}_${v}_$i.err; continue; }
  python -c "
import json
d=json.loads(open('$out/ab_${var}_${v}_$i.log').read().strip().splitlines()[-1]); print('$var=$v', $i, d['value'], d['ms_per_step'], d.get('ms_per_step_median'), 'parity', (d.get('parity') or {}).get('max_rel_diff'))"
done; done
