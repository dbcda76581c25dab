#!/bin/bash
# Collects what profiles/ holds for one build: the default bench line, the rocprofv3 kernel stats of
# the same command, and the PMC passes (one rocprofv3 run per counter group).
# usage (on the GPU box): tools/profile_round.sh <tag>
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
tail -c 600 $out/bench.json; echo
timeout -k 10 400 python3 bench.py --scene cornell > $out/bench_cornell.json 2> $out/bench_cornell.err || { tail -5 $out/bench_cornell.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-serial-roofline > $out/trace_bench.json 2> $out/trace.err || { tail -5 $out/trace.err; exit 1; }
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
# the same with one frame in flight: every launch has the GPU to itself
timeout -k 10 400 python3 bench.py --frames-in-flight 1 --no-cpu-baseline > $out/bench_serial.json 2> $out/bench_serial.err || { tail -5 $out/bench_serial.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_serial -- python3 bench.py --frames-in-flight 1 --no-cpu-baseline > $out/trace_serial_bench.json 2> $out/trace_serial.err || { tail -5 $out/trace_serial.err; exit 1; }
cp $(ls $out/trace_serial/*/*kernel_stats.csv | head -1) $out/kernel_stats_serial.csv
[ -n "${SKIP_PMC:-}" ] && { echo done; exit 0; }
bash tools/pmc_passes.sh $out/pmc --steps 3 --warmup 3 > $out/pmc.log 2>&1 || { tail -5 $out/pmc.log; exit 1; }
bash tools/pmc_passes2.sh $out/pmc2 --steps 3 --warmup 3 >> $out/pmc.log 2>&1 || { tail -5 $out/pmc.log; exit 1; }
python3 tools/pmc_summary.py $out/pmc > $out/pmc_summary.txt
python3 tools/pmc_summary.py $out/pmc2 >> $out/pmc_summary.txt
echo done
