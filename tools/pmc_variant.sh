#!/bin/bash
# Developer tool (GPU box): SQ counters + FETCH/WRITE bytes of k_encode for one library variant.
#   bash tools/pmc_variant.sh <name>     (huffman-gpu_amd/variants/libhvlc_<name>.so, or "product" = libhvlc.so)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
N=$1
if [ "$N" = product ]; then export HVLC_LIB=$PWD/huffman-gpu_amd/libhvlc.so; else export HVLC_LIB=$PWD/huffman-gpu_amd/variants/libhvlc_$N.so; fi
OUT=gpurun_out/pmcvar/$N; rm -rf $OUT; mkdir -p $OUT
B="python3 tools/perf_sweep.py --mib ${MIB:-4096} --bpc 0 --workloads ${WL:-H2.2} --iters 3"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
   --output-format csv -d $OUT/p1 -- $B > $OUT/p1.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM \
   --output-format csv -d $OUT/p2 -- $B > $OUT/p2.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p3 -- $B > $OUT/p3.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p4 -- $B > $OUT/p4.log 2>&1 || exit 1
echo "== $N"; python3 tools/pmc_summary.py $OUT
