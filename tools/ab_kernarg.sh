#!/bin/bash
# alternate default / HIP_FORCE_DEV_KERNARG=1 runs of the fresh-batch bench in one lease
out=gpurun_out/$1; mkdir -p $out
for i in 1 2 3; do
  for mode in default dev; do
    if [ $mode = dev ]; then export HIP_FORCE_DEV_KERNARG=1; else unset HIP_FORCE_DEV_KERNARG; fi
    timeout -k 10 300 python bench.py --gpus 1 --steps 300 --warmup 20 --cpu-seconds 0 --recall-steps 0 --profile-steps 0 > $out/ab_${mode}_$i.log 2> $out/ab_${mode}_$i.err
    python -c "
import json,sys
d=json.loads(open('$out/ab_${mode}_$i.log').read().strip().splitlines()[-1]); print('$mode', $i, d['value'], d['ms_per_step'])"
  done
done
