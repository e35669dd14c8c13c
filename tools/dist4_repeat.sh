#!/bin/bash
# developer tool: the 4-rank staged sharded run (tests/dist_gpu_worker.py), up to $1 times in sequence, stopping at the first failure
n=${1:-4}; out=gpurun_out/${2:-d4}; mkdir -p $out
export MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=4
for i in $(seq 1 $n); do
  mkdir -p $PWD/$out/dbg$i; export SEQREC_DIST_DEBUG=$PWD/$out/dbg$i
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $((29650 + i)) tests/dist_gpu_worker.py > $out/run$i.log 2> $out/run$i.err
  rc=$?; echo "run $i rc=$rc"
  if [ $rc -ne 0 ]; then grep -h "LOSS MISMATCH\|  rank\|AssertionError" $out/run$i.log $out/run$i.err | cut -c1-400; exit 0; fi
done
