#!/bin/bash
# Developer tool: the measurement set behind profiles/ (run on the GPU box from the repo root):
#   ROUND=2 bash tools/measure_round.sh      -> gpurun_out/round/...  (tools/collect_profiles.py copies the summaries)
export TMPDIR=/tmp
O=gpurun_out/round; rm -rf $O; mkdir -p $O
B="python3 bench.py"
timeout -k 10 300 $B > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $B --no-cpu-baseline > $O/prof.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-probe > $O/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-probe > $O/pmc_write.log 2>&1 || exit 1
# SQ counters of the dominant kernel on the bench workload (two passes: 8 SQ slots each)
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
   --output-format csv -d $O/sq1 -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-probe > $O/sq1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM \
   --output-format csv -d $O/sq2 -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-probe > $O/sq2.log 2>&1 || exit 1
for w in H2.2-256MiB H4.0-4GiB H1.0-4GiB; do timeout -k 10 300 $B --workload $w --no-cpu-baseline > $O/bench_$w.json 2>> $O/bench_n1.err || exit 1; done
timeout -k 10 200 python3 tools/bench_next_rows.py > $O/next_rows.json 2>> $O/bench_n1.err || exit 1
timeout -k 10 200 python3 tools/bench_hist.py > $O/hist.json 2>> $O/bench_n1.err
make -s -C huffman-gpu_amd/host && timeout -k 10 200 ./huffman-gpu_amd/host/hvlc_mgpu --gpus 1 --mib 4096 --iters 20 --dist H2.2 --period tests/golden/g1_period.bin --merge 2>/dev/null | grep '^{' > $O/mgpu_n1.json
timeout -k 10 200 python3 tools/bench_host_boundary.py 1024 5 2>/dev/null | grep '^{' > $O/host_boundary.json
(for a in '256 any' '1024 long'; do timeout -k 10 200 python3 tools/slow_path_check.py $a 2>/dev/null | grep -v amdgpu.ids; done) > $O/slow_path.txt
cut -c1-400 $O/bench_n1.json; for w in H2.2-256MiB H4.0-4GiB H1.0-4GiB; do cut -c1-200 $O/bench_$w.json; done; cat $O/next_rows.json $O/mgpu_n1.json
