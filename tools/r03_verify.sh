#!/bin/bash
# Round-3 verification on the GPU box (through gpurun from the repo root).  Outputs under gpurun_out/r03b/.
#   tools/r03_verify.sh kernels   copy-rate shapes, A/B of the step kernels against the previous build, GPU tests
#   tools/r03_verify.sh bench     the N = 1 bench line, the two-rank rehearsal of configs 3/4 with the LSTMs trained on the
#                                 reference's full schedule
out=gpurun_out/r03b
mkdir -p $out
if [ "$1" = kernels ]; then
  tools/ab/copy_bw > $out/copy_bw.txt 2>&1
  python tools/ab.py base=tools/ab/libsaa_base.so new=synchronization_avoiding_algorithms_amd/libsaa_hip.so > $out/ab.txt 2>&1
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/tests.log 2>&1; rc=$?
  echo "pytest rc=$rc" >> $out/tests.log
  exit $rc
fi
timeout -k 10 400 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err || exit 1
timeout -k 10 1000 python bench.py --gpus 2 --same-device --backend gloo --refine 6 --steps 200 --warmup 50 --sa-train-epochs 0 \
  --sa-train-seconds 800 --budget-s 950 > $out/bench_2ranks_full_schedule.json 2> $out/bench_2ranks_full_schedule.err
echo "2-rank rc=$?" >> $out/bench_n1.err
