#!/bin/bash
# One step in flight, all four scenes: tools/all_scenes_ab.sh label=lib.so ...   (ms per step and k_intersect rays/s)
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  for sc in dragon cornell helmet composite; do
    extra=""; [ $sc = composite ] && extra="--width 3840 --height 2160 --spp 64"
    out=$(PT_HIP_LIB=$lib timeout -k 10 200 python bench.py --scene $sc $extra --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --frames-in-flight 1 2>/dev/null | tail -1)
    python3 -c "import json,sys; j=json.loads(sys.argv[3]); print('%-8s %-9s ms/step %7.2f rays/s %.3e intersect %.3e'%(sys.argv[1],sys.argv[2],j['ms_per_step'],j['value'],j['roofline']['intersect_rays_per_s']))" $label $sc "$out"
  done
done
