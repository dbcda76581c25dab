#!/bin/bash
# Two ranks of bench.py on ONE GPU (gloo all-reduce) under a few toggles; one line per case in gpurun_out/reh_diag.txt.
export PT_BENCH_REHEARSAL=1
out=gpurun_out/reh_diag.txt; : > $out
i=0
for spec in "$@"; do
  i=$((i+1)); label=${spec%%:*}; rest=${spec#*:}; envs=${rest%%:*}; args=${rest#*:}
  echo "== $label env[$envs] args[$args]" >> $out
  env $envs timeout -k 10 ${REH_TIMEOUT:-90} python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29520+i)) bench.py --gpus 2 --steps 2 --warmup 1 --no-secondary $args > gpurun_out/reh_$label.out 2> gpurun_out/reh_$label.err
  echo "rc=$?" >> $out
  python3 - gpurun_out/reh_$label.out >> $out 2>&1 <<'PY'
import json,sys
rows=[l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')]
if rows:
    j=json.loads(rows[-1]); k=j['roofline']['per_kernel']
    print('Grays/s %.3f ms/step %.1f k_primary %.2f k_intersect %.2f fif %s' % (j['value']/1e9, j['ms_per_step'], k['k_primary']['avg_launch_ms'], k['k_intersect']['avg_launch_ms'], j['roofline'].get('frames_in_flight')))
else:
    print('no line')
PY
done
cat $out
