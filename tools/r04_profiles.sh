#!/bin/bash
# Round-4 evidence, collected on the GPU box (through gpurun from the repo root); everything lands under gpurun_out/r04/
# and is folded into profiles/ afterwards by tools/r04_fold_profiles.sh.   usage: tools/r04_profiles.sh <part>
#   bench     the driver's command (all legs of the N = 1 line) and the default flags
#   stats     rocprofv3 --kernel-trace --stats of the driver's command (headline + roofline legs)
#   pmc19     PMC passes of the resident kernel on the 1M-tet beam
#   pmc38     PMC passes of the one-launch-per-step kernel on the 8.2M-tet beam
#   pmcdel    PMC passes of the resident kernel on the Delaunay mesh
#   stamps    in-kernel stamps of the resident kernel (diagnostic build)
#   graphs    tools/graph_staleness.py, one changed variable per run
root=$PWD
out=$root/gpurun_out/r04
mkdir -p $out
case "$1" in
bench)
  python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1_driver_flags.json 2> $out/bench_n1_driver_flags.err
  ;;
bench_default)
  python bench.py > $out/bench_n1.json 2> $out/bench_n1.err
  ;;
stats)
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o run --output-format csv -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --legs roofline > $out/stats.log 2>&1; echo "rocprofv3 exit status $?" >> $out/stats.log)
  ;;
pmc19) tools/pmc_collect.sh r04_resident19 19 --warmup 1000 ;;
pmc38) tools/pmc_collect.sh r04_fused38 38 --steps 200 --warmup 50 --no-split-stepping ;;
pmcdel) tools/pmc_collect.sh r04_resident19_delaunay 19 --warmup 1000 --mesh delaunay ;;
stamps)
  python tools/persist_stamps.py 19 --json=$out/resident_stamps.json > $out/resident_stamps.txt 2>&1
  ;;
graphs)
  g=$out/graphs; mkdir -p $g
  run() { name=$1; shift; timeout -k 10 240 env "$@" > $g/$name.txt 2>&1; echo "exit $?" >> $g/$name.txt; }
  run baseline            python tools/graph_staleness.py
  run kernarg_pool_4x     HSA_KERNARG_POOL_SIZE=4194304 python tools/graph_staleness.py
  run kernarg_pool_quarter HSA_KERNARG_POOL_SIZE=262144 python tools/graph_staleness.py
  run packet_capture_off  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tools/graph_staleness.py
  run filler_big          python tools/graph_staleness.py --filler big
  run filler_copy         python tools/graph_staleness.py --filler copy
  run filler_graph        python tools/graph_staleness.py --filler graph
  run zero_mul            python tools/graph_staleness.py --zero mul
  run zero_none           python tools/graph_staleness.py --zero none
  run pass_mse            python tools/graph_staleness.py --pass mse
  run pass_scalar         python tools/graph_staleness.py --pass scalar
  run replay_side         python tools/graph_staleness.py --replay-stream side
  run kernarg_copy_opt_off DEBUG_HIP_KERNARG_COPY_OPT=0 python tools/graph_staleness.py
  run force_dev_kernarg_0 HIP_FORCE_DEV_KERNARG=0 python tools/graph_staleness.py
  run blit_kernarg_opt_off DEBUG_CLR_BLIT_KERNARG_OPT=0 python tools/graph_staleness.py
  ;;
graphs2)
  g=$out/graphs; mkdir -p $g
  run() { name=$1; shift; timeout -k 10 240 env "$@" > $g/$name.txt 2>&1; echo "exit $?" >> $g/$name.txt; }
  run step1               python tools/graph_staleness.py --step 1 --max 40
  run mse_width8          python tools/graph_staleness.py --pass mse --width 8
  run mse_width64         python tools/graph_staleness.py --pass mse --width 64
  run train_graph         python tools/wide_lstm_check.py 9126 12 graph
  run train_graph_nocapture DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tools/wide_lstm_check.py 9126 12 graph
  run train_eager         python tools/wide_lstm_check.py 9126 12 eager
  ;;
esac
