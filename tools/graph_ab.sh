#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
for i in 1 2; do for g in 0 1; do
  SEQREC_SCAN_GRAPH=$g timeout -k 10 300 python tools/host_profile.py plain 600 > $out/g_plain_${g}_$i.log 2>&1; echo "plain graph=$g run $i"; grep "ms/step" $out/g_plain_${g}_$i.log
done; done
