#!/bin/bash
# Collects what profiles/ holds for one build: the default bench line, the rocprofv3 kernel stats of
# the same command and of the one-frame-at-a-time run, the PMC passes (one rocprofv3 run per counter
# group) and the per-configuration lines of BASELINE.md.
# usage (on the GPU box): tools/profile_round.sh <tag>     (SKIP_PMC=1, SKIP_CONFIGS=1 to shorten)
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
run() { local name=$1; shift; timeout -k 10 500 "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -5 $out/$name.err; exit 1; }; }
prof() { local name=$1; shift; timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$name -- "$@" > $out/trace_$name.json 2> $out/trace_$name.err || { echo "trace $name failed"; tail -5 $out/trace_$name.err; exit 1; }
                               cp $(ls $out/trace_$name/*/*kernel_stats.csv | head -1) $out/${name}_kernel_stats.csv; }
run bench python3 bench.py
tail -c 400 $out/bench.json; echo
prof default python3 bench.py --no-serial-roofline --no-secondary --no-cpu-baseline
# one frame at a time: every launch has the GPU to itself
run serial_bench python3 bench.py --frames-in-flight 1 --no-cpu-baseline --no-secondary
prof serial python3 bench.py --frames-in-flight 1 --no-cpu-baseline --no-secondary
if [ -z "${SKIP_CONFIGS:-}" ]; then
  # BASELINE.json's configurations on one GPU (config 4 and 5 at a reduced sample count: they are multi-GPU runs)
  run cfg1_cornell_512 python3 bench.py --scene cornell --width 512 --height 512 --spp 16 --steps 1 --warmup 1 --frames-in-flight 1 --no-secondary
  run cfg2_cornell_1080p python3 bench.py --scene cornell --spp 256 --steps 4 --frames-in-flight 1 --no-secondary
  run cfg2_cube_1080p python3 bench.py --scene cube --spp 256 --steps 4 --frames-in-flight 1 --no-secondary
  run cfg3_helmet_1080p python3 bench.py --scene helmet --total-spp 1024 --frames-in-flight 1 --no-secondary
  run cfg4_dragon_1080p_4096spp python3 bench.py --total-spp 4096 --no-secondary --no-cpu-baseline
  run cfg5_composite_2160p python3 bench.py --scene composite --width 3840 --height 2160 --spp 64 --total-spp 256 --frames-in-flight 1 --no-secondary --cpu-spp 2
  for c in cornell cube helmet composite; do
    case $c in composite) extra="--width 3840 --height 2160 --spp 64";; *) extra="";; esac
    prof cfg_$c python3 bench.py --scene $c $extra --steps 3 --warmup 1 --frames-in-flight 1 --no-cpu-baseline --no-secondary
  done
fi
# lane utilisation of the scheduled traversals (needs variants/stats.so: tools/build_variant.sh stats -DPT_SCHED_STATS)
if [ -f variants/stats.so ]; then
  for c in dragon cornell helmet; do
    SPP=32 PT_HIP_LIB=variants/stats.so PT_HIP_SCHED_MIN_RAYS=0 timeout -k 10 200 python3 tools/sched_stats.py $c > $out/sched_stats_$c.txt 2>&1
  done
  SPP=8 PT_HIP_LIB=variants/stats.so PT_HIP_SCHED_MIN_RAYS=0 timeout -k 10 300 python3 tools/sched_stats.py composite 3840 2160 > $out/sched_stats_composite.txt 2>&1
  bash tools/ab_env.sh "auto:PT_X=0" "sched:PT_HIP_TRAVERSAL=sched" "plain:PT_HIP_TRAVERSAL=plain" > $out/schedules_ab.txt 2>&1
fi
[ -n "${SKIP_PMC:-}" ] && { echo done; exit 0; }
bash tools/pmc_passes.sh $out/pmc --steps 3 --warmup 3 --frames-in-flight 1 > $out/pmc.log 2>&1 || { tail -5 $out/pmc.log; exit 1; }
bash tools/pmc_passes2.sh $out/pmc2 --steps 3 --warmup 3 --frames-in-flight 1 >> $out/pmc.log 2>&1 || { tail -5 $out/pmc.log; exit 1; }
python3 tools/pmc_summary.py $out/pmc > $out/pmc_summary.txt
python3 tools/pmc_summary.py $out/pmc2 >> $out/pmc_summary.txt
python3 tools/pmc_traffic.py $out/pmc --key "dragon 1920x1080 256spp depth10" -- --no-cpu-baseline --steps 3 --warmup 3 --frames-in-flight 1 > $out/traffic_dragon.json
# the other scenes of the metric and of BASELINE's configurations: the four passes the roofline needs
for c in cornell helmet composite cube cornell512; do
  case $c in composite) extra="--width 3840 --height 2160 --spp 64"; key="composite 3840x2160 64spp depth10";;
             cornell512) extra="--width 512 --height 512 --spp 16"; key="cornell 512x512 16spp depth10";;
             *) extra=""; key="$c 1920x1080 256spp depth10";; esac
  scene=${c%512}
  PMC_MIN=1 PMC_TIMEOUT=200 bash tools/pmc_passes.sh $out/pmc_$c --scene $scene $extra --steps 2 --warmup 2 --frames-in-flight 1 >> $out/pmc.log 2>&1 || { tail -5 $out/pmc.log; exit 1; }
  python3 tools/pmc_traffic.py $out/pmc_$c --key "$key" -- --scene $scene $extra --no-cpu-baseline --steps 2 --warmup 2 --frames-in-flight 1 > $out/traffic_$c.json
  python3 tools/pmc_summary.py $out/pmc_$c > $out/pmc_summary_$c.txt
done
echo done
