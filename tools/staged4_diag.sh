#!/bin/bash
# developer tool (round 3): the staged 4-rank worker with the failure report on (tests/dist_gpu_worker.py prints
# ShardedEngine.failure_report() when Engine.check_status() raises) -- at most $1 runs, stops at the first failure
n=${1:-6}; out=gpurun_out/${2:-s4}; mkdir -p $out
export MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=4
for i in $(seq 1 $n); do
  timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $((29700 + i)) tests/dist_gpu_worker.py > $out/run$i.log 2> $out/run$i.err
  rc=$?; echo "run $i rc=$rc"
  if [ $rc -ne 0 ]; then grep -h "FAILURE REPORT" $out/run$i.log | cut -c1-3000; exit 0; fi
done
