#!/bin/bash
# One, two and two prioritised steps in flight, alternated: tools/fif_priority.sh [rounds]
for r in $(seq 1 ${1:-3}); do
for spec in ${SPECS:-"fif1:A=1:1" "fif2:A=1:2" "fif2_prio:PT_HIP_LANE_PRIORITY=1:2"}; do
  label=${spec%%:*}; rest=${spec#*:}; envs=${rest%%:*}; f=${rest#*:}
  out=$(env $envs timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-serial-roofline --frames-in-flight $f ${EXTRA:-} 2>/dev/null | tail -1)
  python3 -c "
import json,sys; j=json.loads(sys.argv[2]); print('%-12s ms/step %7.2f rays/s %.3e' % (sys.argv[1], j['ms_per_step'], j['value']))" "$label" "$out"
done
done
