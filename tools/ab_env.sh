#!/bin/bash
# usage: tools/ab_env.sh VAR "v1 v2 ..." [bench args...]   -- run bench.py once per value of VAR, print the headline numbers
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v python bench.py --recall-steps 0 --cpu-seconds 0 "$@" 2>/dev/null > /tmp/ab_$$.json || exit 1
  python - "$var=$v" /tmp/ab_$$.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); k = d["kernels"]
print(sys.argv[1], d["value"], d["ms_per_step"], k.get("seqrec_rnn_fwd_stepwise", {}).get("avg_us"), k.get("seqrec_rnn_bwd_stepwise", {}).get("avg_us"))
PY
done
