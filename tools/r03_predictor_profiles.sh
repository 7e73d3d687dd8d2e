#!/bin/bash
# Profiles of the predictor kernels (run on the GPU box from the repo root): window times against the PyTorch-ROCm route,
# rocprofv3 kernel stats of the same command, in-kernel stamps of the input-projection GEMM.
# (tools/ab/gemm_stamps must have been built here: hipcc -O3 --offload-arch=gfx950 -std=c++17 -DSAA_GEMM_STAMPS
#  tools/gemm_stamps.hip -o tools/ab/gemm_stamps)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r03_predictor
mkdir -p $out
timeout -k 10 200 python3 $root/tools/predictor_point.py 24 1000 9126 > $out/point.txt 2>&1 || echo "point failed rc=$?" >> $out/failed.txt
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $out/stats -o run --output-format csv -- python3 $root/tools/predictor_point.py 9126 --native-only > $out/stats.log 2>&1; echo "rocprofv3 exit status $?" >> $out/stats.log)
timeout -k 10 100 $root/tools/ab/gemm_stamps > $out/gemm_stamps.txt 2>&1 || echo "stamps failed rc=$?" >> $out/failed.txt
grep "saa::" $out/stats/run_kernel_stats.csv
cat $out/point.txt
