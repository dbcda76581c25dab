#!/bin/bash
# Per-bounce kernel durations of the bench workload from a rocprofv3 kernel trace.
# usage: tools/depth_times.sh <label> [lib.so]   (run on the GPU box)
label=$1; lib=${2:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/depth_$label
[ -n "$lib" ] && export PT_HIP_LIB=$lib
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --spp ${SPP:-16} --frames-in-flight 1 ${EXTRA:-} > $out.log 2>&1 || { tail -3 $out.log; exit 1; }
python3 - "$label" $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[2] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
frames, cur = [], None
for r in rows:
    n = r['Kernel_Name']; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if 'k_init_counts' in n: cur = {'I': [], 'S': [], 't0': int(r['Start_Timestamp'])}; frames.append(cur)
    elif cur is not None and ('k_intersect' in n or 'k_primary' in n): cur['I'].append(d)
    elif cur is not None and 'k_shade' in n: cur['S'].append(d)
    if cur is not None: cur['t1'] = int(r['End_Timestamp'])
fr = frames[-4:]
for k in 'IS':
    per = [round(sum(f[k][d] for f in fr) / len(fr)) for d in range(len(fr[0][k]))]
    print(sys.argv[1], k, per, 'sum', sum(per))
print(sys.argv[1], 'frame span us', [round((f['t1'] - f['t0']) / 1e3) for f in fr])
PY
