#!/bin/bash
# usage: helmet_ab.sh label=lib ...   helmet/composite/dragon one step at a time
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  for sc in helmet dragon; do
    out=$(PT_HIP_LIB=$lib timeout -k 10 200 python bench.py --scene $sc --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --frames-in-flight 1 2>/dev/null | tail -1)
    python3 -c "import json,sys; j=json.loads(sys.argv[3]); print('%-8s %-9s ms/step %7.2f rays/s %.3e intersect %.3e'%(sys.argv[1],sys.argv[2],j['ms_per_step'],j['value'],j['roofline']['intersect_rays_per_s']))" $label $sc "$out"
  done
done
