#!/bin/bash
# A/B of environment settings (and library builds) in one GPU session, one frame at a time:
# tools/ab_env.sh "<label>:<ENV=VAL ...>" ...   prints ms per step and the traversal kernels' launch averages
for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}
  out=$(env $envs timeout -k 10 200 python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-secondary --frames-in-flight 1 --spp ${SPP:-256} ${EXTRA:-} 2>/dev/null | tail -1)
  python3 - "$label" "$out" <<'PY'
import json,sys
j=json.loads(sys.argv[2]); r=j["roofline"]["per_kernel"]
print("%-16s ms/step %7.2f  rays/s %.3e | k_primary %7.2f ms  k_intersect %6.3f ms"%(sys.argv[1], j["ms_per_step"], j["value"], r["k_primary"]["avg_launch_ms"], r["k_intersect"]["avg_launch_ms"]))
PY
done
