#!/bin/bash
# Developer tool: the measurement set behind profiles/ (run on the GPU box from the repo root).
export TMPDIR=/tmp
O=gpurun_out/round; rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1 || exit 1
for w in H2.2-256MiB H4.0-4GiB H1.0-4GiB; do timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>> $O/bench_n1.err || exit 1; done
cat $O/bench_n1.json; for w in H2.2-256MiB H4.0-4GiB H1.0-4GiB; do cut -c1-200 $O/bench_$w.json; done
