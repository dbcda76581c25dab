#!/bin/bash
# Bench line summary for combinations of frames in flight and samples per step: tools/lanes_bench.sh "<lanes...>" "<spp...>"
for spp in $2; do for l in $1; do
  out=$(timeout -k 10 200 python bench.py --frames-in-flight $l --steps $((128 / spp > 4 ? 128 / spp : 4)) --warmup 2 --no-cpu-baseline --no-secondary --spp $spp 2>/dev/null | tail -1)
  python3 - "$l" "$spp" "$out" <<'PY'
import json,sys
j=json.loads(sys.argv[3])
print("lanes %s spp %3s: rays/s %.3e ms/16spp %6.2f | intersect rays/s %.3e frac %.3f"%(sys.argv[1],sys.argv[2],j["value"],j["ms_per_step"]*16/int(sys.argv[2]),j["roofline"]["intersect_rays_per_s"],j["roofline"]["frac"]))
PY
done; done
