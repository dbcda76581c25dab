#!/bin/bash
# Per-kernel average durations of the bench workload for one library build (run on the GPU box).
# usage: tools/kernel_times.sh <label> [lib.so]
label=$1; lib=${2:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/kt_$label
[ -n "$lib" ] && export PT_HIP_LIB=$lib
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --frames-in-flight 1 > $out.log 2>&1 || { tail -3 $out.log; exit 1; }
python3 - "$label" $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[2] + '/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if any(k in n for k in ('k_intersect', 'k_shade', 'k_accumulate', 'k_primary')):
        print('%-6s %-60s calls %4s avg %9.1f us' % (sys.argv[1], n[:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
grep '^{"metric' $out.log | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(sys.argv[1], 'ms/step', round(j['ms_per_step'],2), 'rays/s %.3e' % j['value'])" $label
