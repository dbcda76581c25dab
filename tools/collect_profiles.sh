#!/bin/bash
# Copies what tools/profile_round.sh <tag> left under gpurun_out/<tag>/ into profiles/ under the names the
# tests and documents use, and merges the round's PMC records into profiles/traffic.json.
# usage (in the repository, after the GPU call has merged its files back): tools/collect_profiles.sh <tag>
set -eu
tag=$1
cd "$(dirname "$0")/.."
src=gpurun_out/$tag
cp $src/bench.json profiles/${tag}_bench.json
cp $src/default_kernel_stats.csv profiles/${tag}_kernel_stats.csv
cp $src/serial_bench.json profiles/${tag}_serial_bench.json
cp $src/serial_kernel_stats.csv profiles/${tag}_serial_kernel_stats.csv
for f in $src/cfg[0-9]*.json; do cp $f profiles/${tag}_$(basename $f); done
for f in $src/cfg_*_kernel_stats.csv; do cp $f profiles/${tag}_$(basename $f); done
cp $src/pmc_summary.txt profiles/${tag}_pmc.txt
for f in $src/pmc_summary_*.txt; do n=$(basename $f .txt); cp $f profiles/${tag}_pmc_${n#pmc_summary_}.txt; done
for f in $src/sched_stats_*.txt $src/schedules_ab.txt; do [ -f $f ] && cp $f profiles/${tag}_$(basename $f); done
python3 tools/merge_traffic.py $src
