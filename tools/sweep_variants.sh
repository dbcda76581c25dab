#!/bin/bash
# tools/sweep_variants.sh <name>...: one frame at a time (least run-to-run noise), ms per step for variants/<name>.so and the tree's library
for n in base "$@" base; do
  lib=variants/$n.so; [ "$n" = base ] && lib=path-tracer_amd/lib/libpt_hip.so
  out=$(PT_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --frames-in-flight 1 --spp ${SPP:-64} 2>/dev/null | tail -1)
  python3 -c "import json,sys; j=json.loads(sys.argv[2]); print('%-10s ms/step %6.2f  rays/s %.3e'%(sys.argv[1], j['ms_per_step'], j['value']))" "$n" "$out"
done
