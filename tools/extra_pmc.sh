#!/bin/bash
# the PMC passes of the two configurations added after the first profile round (cube 1080p, Cornell 512x512 x 16 spp)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_v1c; mkdir -p $out
for c in cube cornell512; do
  case $c in cornell512) extra="--width 512 --height 512 --spp 16"; key="cornell 512x512 16spp depth10";; *) extra=""; key="$c 1920x1080 256spp depth10";; esac
  scene=${c%512}
  PMC_MIN=1 PMC_TIMEOUT=200 bash tools/pmc_passes.sh $out/pmc_$c --scene $scene $extra --steps 2 --warmup 2 --frames-in-flight 1 >> $out/pmc.log 2>&1
  python3 tools/pmc_traffic.py $out/pmc_$c --key "$key" -- --scene $scene $extra --no-cpu-baseline --steps 2 --warmup 2 --frames-in-flight 1 > $out/traffic_$c.json
  python3 tools/pmc_summary.py $out/pmc_$c > $out/pmc_summary_$c.txt
done
echo done
