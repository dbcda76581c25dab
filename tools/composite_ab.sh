#!/bin/bash
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  out=$(PT_HIP_LIB=$lib timeout -k 10 200 python bench.py --scene composite --width 3840 --height 2160 --spp 64 --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --frames-in-flight 1 2>/dev/null | tail -1)
  python3 -c "import json,sys; j=json.loads(sys.argv[2]); print('%-8s composite ms/step %7.2f rays/s %.3e intersect %.3e'%(sys.argv[1],j['ms_per_step'],j['value'],j['roofline']['intersect_rays_per_s']))" $label "$out"
  out=$(PT_HIP_LIB=$lib timeout -k 10 200 python bench.py --scene cornell --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --frames-in-flight 1 2>/dev/null | tail -1)
  python3 -c "import json,sys; j=json.loads(sys.argv[2]); print('%-8s cornell   ms/step %7.2f rays/s %.3e intersect %.3e'%(sys.argv[1],j['ms_per_step'],j['value'],j['roofline']['intersect_rays_per_s']))" $label "$out"
done
