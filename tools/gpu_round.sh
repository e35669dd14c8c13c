#!/bin/bash
# usage: tools/gpu_round.sh <tag> [steps...] : steps among tests bench stall trace resident ; stops after a timeout.
tag=$1; shift; steps=${@:-tests bench}
out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; timeout -k 10 $TMO "$@" > $out/$name.log 2> $out/$name.err; rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi; }
for s in $steps; do
case $s in
tests) TMO=900 run tests python -m pytest tests -m gpu -q -x; tail -5 $out/tests.log;;
testsall) TMO=900 run tests python -m pytest tests -m gpu -q; tail -15 $out/tests.log;;
bench) TMO=400 run bench_driver python bench.py --gpus 1 --steps 20 --warmup 5; tail -c 1500 $out/bench_driver.log;;
resident) TMO=400 run bench_resident python bench.py --gpus 1 --steps 200 --warmup 20 --resident --cpu-seconds 0 --recall-steps 0; tail -c 600 $out/bench_resident.log;;
fresh200) TMO=400 run bench_fresh200 python bench.py --gpus 1 --steps 200 --warmup 20 --cpu-seconds 0 --recall-steps 0; tail -c 600 $out/bench_fresh200.log;;
stall) TMO=200 run stall python tools/stall_probe.py 12; cat $out/stall.log; TMO=200 run stall_freeze python tools/stall_probe.py 12 freeze; cat $out/stall_freeze.log;;
graph) SEQREC_SCAN_GRAPH=1 TMO=400 run bench_graph python bench.py --gpus 1 --steps 200 --warmup 20 --cpu-seconds 0 --recall-steps 0; tail -c 300 $out/bench_graph.log;;
dbgtopk) TMO=300 run dbgtopk python tools/debug_topk.py c5; cat $out/dbgtopk.log;;
trace) root=$PWD; cd /tmp; export TMPDIR=/tmp
   timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $root/$out/trace -- python3 $root/tools/stall_probe.py 8 > $root/$out/trace.log 2>&1; echo "trace rc=$?"
   cd $root; find $out/trace -name '*.csv' | head; python tools/trace_stall.py $out/trace > $out/trace_summary.txt 2>&1; cat $out/trace_summary.txt; find $out/trace -name '*.csv' -size +20M -delete;;
esac
done
