#!/bin/bash
# Developer tool (GPU box): a wider set of SQ / SQC counters of k_encode for one library (instruction cache, LDS queues).
#   bash tools/pmc_deep.sh <name>      ("product" = libhvlc.so, else variants/libhvlc_<name>.so)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
N=$1
if [ "$N" = product ]; then export HVLC_LIB=$PWD/huffman-gpu_amd/libhvlc.so; else export HVLC_LIB=$PWD/huffman-gpu_amd/variants/libhvlc_$N.so; fi
OUT=gpurun_out/pmcdeep/$N; rm -rf $OUT; mkdir -p $OUT
B="python3 tools/perf_sweep.py --mib ${MIB:-4096} --bpc 0 --workloads ${WL:-H2.2} --iters 3"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG" \
           "SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $B > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
echo "== $N"; python3 tools/pmc_summary.py $OUT
