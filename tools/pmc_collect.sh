#!/bin/bash
# Collects the PMC passes of one bench command on the GPU box (run through gpurun from the repo root):
#   tools/pmc_collect.sh <key> <refine n> [extra bench flags]
# One rocprofv3 --pmc pass per counter group (FETCH_SIZE and WRITE_SIZE each in a pass of their own, as
# MI355X_MICROARCH.md "HBM" prescribes; --pmc is never combined with tracing flags), results under gpurun_out/pmc_<key>/.
key=$1; n=$2; shift 2
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$PWD
out=$root/gpurun_out/pmc_$key
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $grp -d $out/g$i -o run --output-format csv -- python3 $root/bench.py --steps 1000 --warmup 100 --legs none --min-timed-ms 50 --refine $n "$@" > $out/g$i.log 2>&1 || echo "group $i failed rc=$?" >> $out/failed.txt
done
