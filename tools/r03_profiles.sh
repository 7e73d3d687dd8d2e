#!/bin/bash
# Round-3 evidence, collected on the GPU box (through gpurun from the repo root); everything lands under gpurun_out/r03_final/
# and is folded into profiles/ afterwards by tools/r03_fold_profiles.sh.   usage: tools/r03_profiles.sh <part>
#   bench   bench lines: default flags, the driver's flags, 8.2M tets on one GPU, the jittered mesh
#   stats   rocprofv3 --kernel-trace --stats of the driver's command
#   pmc19   PMC passes of the resident kernel on the 1M-tet beam (FETCH_SIZE / WRITE_SIZE / SQ groups / fp64 instruction counts)
#   pmc38   PMC passes of the one-launch-per-step kernel on the 8.2M-tet beam
#   stamps  in-kernel stamps of both kernels (diagnostic builds)
root=$PWD
out=$root/gpurun_out/r03_final
mkdir -p $out
case "$1" in
bench)
  python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1_driver_flags.json 2> $out/bench_n1_driver_flags.err
  python bench.py > $out/bench_n1.json 2> $out/bench_n1.err
  python bench.py --steps 200 --warmup 50 --refine 38 --no-cpu-baseline > $out/bench_n1_8Mtets.json 2> $out/bench_8m.err
  python bench.py --mesh jittered --no-cpu-baseline > $out/bench_n1_jittered.json 2> $out/bench_jittered.err
  python tools/unstructured_point.py > $out/unstructured_1M_tets.txt 2>&1
  ;;
stats)
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o run --output-format csv -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/stats.log 2>&1; echo "rocprofv3 exit status $?" >> $out/stats.log)
  rocprofv3 -L 2>/dev/null | grep -i "F64\|FLOPS" > $out/counters_f64.txt
  ;;
pmc19) tools/pmc_collect.sh r03_resident19 19 --warmup 1000 ;;
pmc38) tools/pmc_collect.sh r03_fused38 38 --steps 200 --warmup 50 ;;
stamps)
  python tools/persist_stamps.py 19 --json=$out/resident_stamps.json > $out/resident_stamps.txt 2>&1
  python tools/stamps.py 38 > $out/fused_8Mtets_stamps.txt 2>&1
  python tools/ablate.py 38 > $out/fused_8Mtets_ablation.txt 2>&1
  ;;
esac
