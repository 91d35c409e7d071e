#!/bin/bash
# Developer tool: SQ counters of k_decode (product build), two PMC passes over tools/bench_next_rows.py.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmcdec; rm -rf $OUT; mkdir -p $OUT
B="python3 tools/bench_next_rows.py ${MIB:-1024}"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
   --output-format csv -d $OUT/p1 -- $B > $OUT/p1.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM \
   --output-format csv -d $OUT/p2 -- $B > $OUT/p2.log 2>&1 || exit 1
KERNEL=${KERNEL:-k_decode} python3 tools/pmc_summary.py $OUT
