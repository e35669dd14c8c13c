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
prof) root=$PWD; cd /tmp; export TMPDIR=/tmp; P=$root/$out/prof; mkdir -p $P
   timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/ks -o ks -- python3 $root/bench.py --steps 40 --warmup 5 --cpu-seconds 0 --recall-steps 0 --profile-steps 0 > $P/bench_under_rocprof.json 2> $P/ks.err; echo "ks rc=$?"
   timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/w -o w -- python3 $root/tools/stall_probe.py 2 > $P/w.log 2>&1; echo "w rc=$?"
   timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/f -o f -- python3 $root/tools/stall_probe.py 2 > $P/f.log 2>&1; echo "f rc=$?"
   timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/wr -o wr -- python3 $root/tools/stall_probe.py 2 > $P/wr.log 2>&1; echo "wr rc=$?"
   timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/m -o m -- python3 $root/tools/stall_probe.py 2 > $P/m.log 2>&1; echo "m rc=$?"
   cd $root; find $P -name '*.csv' | head -20
   python tools/pmc_widths.py $(find $P/w -name '*counter_collection.csv' | head -1) $P/scan_widths.json > $P/scan_widths.txt 2>&1; cat $P/scan_widths.txt
   python tools/pmc_traffic.py $(find $P/f -name '*counter_collection.csv' | head -1) $(find $P/wr -name '*counter_collection.csv' | head -1) $P/pmc_hbm_traffic.json
   python tools/pmc_mfma.py $(find $P/m -name '*counter_collection.csv' | head -1) $P/pmc_mfma.json | head -30
   find $P -name '*kernel_stats.csv' -exec cp {} $P/kernel_stats.csv \;
   find $P -name '*.csv' -size +3M -delete;;
kernarg) HIP_FORCE_DEV_KERNARG=1 TMO=400 run bench_kernarg1 python bench.py --gpus 1 --steps 200 --warmup 20 --cpu-seconds 0 --recall-steps 0; tail -c 300 $out/bench_kernarg1.log
   HIP_FORCE_DEV_KERNARG=0 TMO=400 run bench_kernarg0 python bench.py --gpus 1 --steps 200 --warmup 20 --cpu-seconds 0 --recall-steps 0; tail -c 300 $out/bench_kernarg0.log;;
sharded) TMO=400 run bench_sharded python bench.py --gpus 1 --steps 200 --warmup 20 --cpu-seconds 0 --recall-steps 200 --force-sharded; tail -c 400 $out/bench_sharded.log;;
hostprof) TMO=300 run hostprof python tools/host_profile.py; head -50 $out/hostprof.log; TMO=300 run hostprof_sh python tools/host_profile.py sharded; head -60 $out/hostprof_sh.log;;
others) for c in c2 c4 c5; do TMO=400 run bench_$c python bench.py --config $c --steps 200 --warmup 20 --cpu-seconds 0 --recall-steps 300; python -c "
import json; d=json.loads(open('$out/bench_$c.log').read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'], d['recall_at_20'], d['config']['scan'])"; done
   TMO=400 run bench_sorted python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --recall-steps 0 --merge sorted; python -c "
import json; d=json.loads(open('$out/bench_sorted.log').read().strip().splitlines()[-1]); print('sorted', d['value'], d['ms_per_step'], [(k,v['avg_us']) for k,v in d['kernels'].items() if 'merge' in k or 'scatter' in k or 'sqnorm' in k])"
   TMO=400 run bench_sat python bench.py --saturated --steps 40 --warmup 5 --cpu-seconds 0 --recall-steps 0 --train-sessions 40000; python -c "
import json; d=json.loads(open('$out/bench_sat.log').read().strip().splitlines()[-1]); print('saturated', d['value'], d['ms_per_step'], d['tokens_per_s'], [(k[7:],v.get('frac')) for k,v in d['kernels'].items() if v.get('frac')])";;
arb) for c in c4 c5 c3; do TMO=700 run bench_arb_$c python bench.py --config $c --steps 20 --warmup 5 --arbiter on; python -c "
import json; d=json.loads(open('$out/bench_arb_$c.log').read().strip().splitlines()[-1]); p=d['parity']; print('$c', d['value'], p['ok'], p.get('ok_reason'), p['rel_diff_per_step'], p.get('vs_fp64'), {k: p['resync'][k] for k in ('loss_rel_gpu','loss_rel_cpu32','clip_scale_rel_gpu','tokens_ce_clipped','ok','why_not')}); print({k: {f: u[f] for f in ('gpu_l2','cpu32_l2','gpu_flips','cpu32_flips','gpu_l2_rest','cpu32_l2_rest')} for k, u in p['resync']['update'].items()}); print(d.get('notes'))"; done;;
launch2) SEQREC_BENCH_BACKEND=gloo-staged TMO=400 run bench_launch2 python bench.py --gpus 2 --steps 10 --warmup 3 --config c2 --recall-steps 0 --train-sessions 20000 --test-sessions 2000; tail -c 700 $out/bench_launch2.log; tail -5 $out/bench_launch2.err;;
dbgtopk) TMO=300 run dbgtopk python tools/debug_topk.py c5; cat $out/dbgtopk.log;;
trace) root=$PWD; cd /tmp; export TMPDIR=/tmp
   timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $root/$out/trace -- python3 $root/tools/stall_probe.py 8 > $root/$out/trace.log 2>&1; echo "trace rc=$?"
   cd $root; find $out/trace -name '*.csv' | head; python tools/trace_stall.py $out/trace > $out/trace_summary.txt 2>&1; cat $out/trace_summary.txt; find $out/trace -name '*.csv' -size +20M -delete;;
esac
done
