#!/bin/bash
# A/B of library builds in one GPU session: tools/ab_bench.sh <label=lib.so[,ENV=VAL]>... ; prints one line per variant
for spec in "$@"; do
  label=${spec%%=*}; rest=${spec#*=}
  lib=${rest%%,*}; envs=""
  if [[ "$rest" == *,* ]]; then envs=${rest#*,}; envs=${envs//,/ }; fi
  out=$(env PT_HIP_LIB=$lib $envs timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --secondary 2>/dev/null | tail -1)
  python3 - "$label" "$out" <<'PY'
import json,sys
j=json.loads(sys.argv[2])
print("%-14s dragon rays/s %.3e paths/s %.3e ms/step %6.2f | intersect rays/s %.3e | cornell rays/s %.3e"%(sys.argv[1],j["value"],j["paths_per_s"],j["ms_per_step"],j["roofline"]["intersect_rays_per_s"],j.get("cornell",{}).get("rays_per_s",0)))
PY
done
