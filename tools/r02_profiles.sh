#!/bin/bash
# Round-2 evidence, collected on the GPU box in one go (run through gpurun from the repo root); everything lands under
# gpurun_out/r02_final/ and is copied into profiles/ afterwards.
root=$PWD
out=$root/gpurun_out/r02_final
mkdir -p $out
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1_driver_flags.json 2> $out/bench_n1_driver_flags.err
python bench.py > $out/bench_n1.json 2> $out/bench_n1.err
python bench.py --steps 200 --warmup 50 --refine 38 --no-cpu-baseline > $out/bench_n1_8Mtets.json 2> $out/bench_8m.err
python bench.py --gpus 2 --same-device --backend gloo --refine 6 --steps 200 --warmup 50 --sa-train-seconds 25 > $out/bench_2ranks_same_device.json 2> $out/bench_2ranks.err
python tools/peer_loopback.py > $out/peer_loopback.txt 2>&1
python tools/persist_stamps.py > $out/resident_stamps.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o run --output-format csv -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/stats.log 2>&1; echo "rocprofv3 exit status $?" >> $out/stats.log)
tools/pmc_collect.sh r02_resident19 19 --warmup 1000
tools/pmc_collect.sh r02_fused38 38 --steps 200 --warmup 50
