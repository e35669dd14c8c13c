#!/bin/bash
# developer tool: c4 bench under a list of environment settings (one per argument, "NAME=VALUE[,NAME=VALUE]" or "base")
out=gpurun_out/${OUT:-abc4}; mkdir -p $out
for spec in "$@"; do
  tag=$(echo "$spec" | tr ',=' '__')
  ( if [ "$spec" != base ]; then IFS=,; for kv in $spec; do export "$kv"; done; fi
    timeout -k 10 200 python bench.py --config ${CFG:-c4} --steps 40 --warmup 10 --recall-steps 0 --cpu-seconds 0 --tune-steps 0 > $out/$tag.json 2> $out/$tag.err )
  python - "$out/$tag.json" "$spec" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    ks=d["kernels"]
    print("%-44s %9.0f sess/s %.4f ms (median %.4f)  " % (sys.argv[2], d["value"], d["ms_per_step"], d["ms_per_step_median"]) + " ".join("%s=%.0f" % (k.replace("seqrec_","").replace("gemm_f32","g")[:22], v["avg_us"]) for k,v in list(ks.items())[:9]))
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
done
