#!/bin/bash
# Steps in flight against ms per step for one scene: tools/fif_scene.sh <scene> [bench args...]
sc=$1; shift
for f in 1 2 3 4; do
  out=$(timeout -k 10 300 python bench.py --scene $sc --steps 8 --warmup 4 --no-cpu-baseline --no-secondary --no-serial-roofline --frames-in-flight $f "$@" 2>/dev/null | tail -1)
  python3 -c "import json,sys; j=json.loads(sys.argv[3]); print('%-9s in flight %s ms/step %7.2f rays/s %.3e'%(sys.argv[1],sys.argv[2],j['ms_per_step'],j['value']))" $sc $f "$out"
done
