#!/bin/bash
# tools/depth_times.sh for the Cornell scene (run on the GPU box).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/depth_cornell
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --scene cornell --steps 4 --warmup 1 --no-cpu-baseline --spp 64 --frames-in-flight 1 > $out.log 2>&1 || { tail -3 $out.log; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
frames, cur = [], None
for r in rows:
    n = r['Kernel_Name']; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if 'k_init_counts' in n: cur = {'I': [], 'S': []}; frames.append(cur)
    elif cur is not None and ('k_intersect' in n or 'k_primary' in n): cur['I'].append(d)
    elif cur is not None and 'k_shade' in n: cur['S'].append(d)
fr = frames[-4:]
for k in 'IS':
    per = [round(sum(f[k][d] for f in fr) / len(fr)) for d in range(len(fr[0][k]))]
    print('cornell', k, per, 'sum', sum(per))
PY
