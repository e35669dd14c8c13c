#!/bin/bash
# A/B of environment settings on the fresh-batch bench, non-GEMM non-scan kernels shown: ab_env_misc.sh <tag> "VAR=val ..." ...
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
i=0
for rep in 1 2; do for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout -k 10 300 python bench.py --steps 300 --warmup 20 --cpu-seconds 0 --recall-steps 0 --tune-steps 0 > $out/b$i.log 2> $out/b$i.err || { echo "run $i failed"; tail -3 $out/b$i.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('$out/b$i.log').read().strip().splitlines()[-1])
ks=d['kernels']
print('$cfg', $rep, d['value'], d['ms_per_step'], [(k.replace('seqrec_',''), v['avg_us']) for k,v in ks.items() if 'gemm' not in k and 'rnn_' not in k])
PY
done; done
