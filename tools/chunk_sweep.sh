#!/bin/bash
# Throughput of the 256-spp step traced in pixel chunks of <max paths> on <lanes> internal streams (run on the GPU box).
# usage: tools/chunk_sweep.sh "<lanes...>" "<max_paths...>"   (max_paths 0 = the whole step in one chunk)
for mp in $2; do for l in $1; do
  out=$(timeout -k 10 200 python bench.py --frames-in-flight $l --max-paths $mp --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-serial-roofline 2>/dev/null | tail -1)
  python3 - "$l" "$mp" "$out" <<'PY'
import json,sys
j=json.loads(sys.argv[3])
mp=int(sys.argv[2]); paths=mp if mp else 1920*1080*256
print("lanes %s chunk %10d paths (%5.1f GB of queues): rays/s %.3e ms/step %6.2f"%(sys.argv[1],paths,int(sys.argv[1])*paths*128/1e9,j["value"],j["ms_per_step"]))
PY
done; done
