#!/bin/bash
# Would two lock-step groups on separate HIP queues fill each other's tails?  Two PROCESSES (own weights, own queues) decode u users each at the
# same time; compare with one process decoding u and 2u users.  usage: tools/concurrent_groups.sh "2 8" [reps]
reps=${2:-12}
for u in ${1:-2 8}; do
  echo "--- one process, $u users:";        timeout -k 10 200 python tools/batch_run.py $u bssd $reps 2>&1 | grep MARK
  echo "--- one process, $((2*u)) users:";  timeout -k 10 200 python tools/batch_run.py $((2*u)) bssd $reps 2>&1 | grep MARK
  echo "--- two processes, $u users each, concurrently:"
  timeout -k 10 300 python tools/batch_run.py $u bssd $((3*reps)) > gpurun_out/cg_a.log 2>&1 &
  pa=$!
  timeout -k 10 300 python tools/batch_run.py $u bssd $((3*reps)) > gpurun_out/cg_b.log 2>&1 &
  pb=$!
  wait $pa; wait $pb
  grep MARK gpurun_out/cg_a.log gpurun_out/cg_b.log
done
