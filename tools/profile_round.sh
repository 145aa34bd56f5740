#!/bin/bash
# Measurements behind profiles/: bench lines of every workload, rocprofv3 kernel stats of
# the default bench command, and PMC passes per workload (SQ, FETCH_SIZE, WRITE_SIZE: one run
# each, counters only; they come first, the bench lines quote the HBM traffic they measure).  Run on the GPU box from the repo root; results go to
# gpurun_out/round/.   usage: tools/profile_round.sh [workloads...]
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/round
mkdir -p $O
W="${@:-buddha bunny dino_spheres multi default default_500}"
cd /tmp
for w in $W; do
    echo "pmc sq $w"
    timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $O/pmc_sq_$w --output-format csv -- python3 $R/tools/kernel_times.py $w --frames 2 > $O/pmc_sq_$w.log 2>&1 || exit 1
    echo "pmc fetch $w"
    timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_$w --output-format csv -- python3 $R/tools/kernel_times.py $w --frames 2 > $O/pmc_fetch_$w.log 2>&1 || exit 1
    echo "pmc write $w"
    timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_$w --output-format csv -- python3 $R/tools/kernel_times.py $w --frames 2 > $O/pmc_write_$w.log 2>&1 || exit 1
done
cd $R
for w in $W; do
    python tools/pmc_summary.py $O/pmc_sq_$w > $O/pmc_sq_$w.txt
    python tools/pmc_summary.py $O/pmc_fetch_$w > $O/pmc_fetch_$w.txt
    python tools/pmc_summary.py $O/pmc_write_$w > $O/pmc_write_$w.txt
done
# the bench lines take roofline.traffic from profiles/hbm_traffic.json: refresh it first
python tools/hbm_traffic.py $O profiles/hbm_traffic.json && cp profiles/hbm_traffic.json $O/hbm_traffic.json
for w in $W; do
    echo "bench $w"
    timeout -k 10 300 python -u bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err || exit 1
done
cd /tmp
echo "rocprofv3 kernel stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --host-steps 0 > $O/bench_under_rocprof.json 2> $O/stats.err || exit 1
echo done
