#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
for mode in plain sharded; do for g in 0 1; do
  SEQREC_SCAN_GRAPH=$g timeout -k 10 300 python tools/host_profile.py $mode 800 > $out/g_${mode}_$g.log 2>&1; echo "$mode graph=$g"; grep "ms/step" $out/g_${mode}_$g.log
done; done
