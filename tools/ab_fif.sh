#!/bin/bash
# Steps in flight: tools/ab_fif.sh "<label>:<bench args>" ...   prints ms per step for 1, 2 and 3 frames in flight
for spec in "$@"; do
  label=${spec%%:*}; args=${spec#*:}
  for f in 1 2 3; do
    out=$(timeout -k 10 300 python bench.py --steps ${STEPS:-6} --warmup 3 --no-cpu-baseline --no-secondary --no-serial-roofline --frames-in-flight $f $args 2>/dev/null | tail -1)
    python3 - "$label fif=$f" "$out" <<'PY'
import json,sys
j=json.loads(sys.argv[2])
print("%-28s ms/step %7.2f  rays/s %.3e  paths/s %.3e"%(sys.argv[1], j["ms_per_step"], j["value"], j["paths_per_s"]))
PY
  done
done
