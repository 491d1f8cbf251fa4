#!/bin/bash
# where a closure's solve goes in the non-lifelong replay: host preparation (KH_SPA_DEBUG lines) against the rest
KH_SPA_DEBUG=1 python tools/replay.py --scans 3000 --no-lifelong > /tmp/rs.json 2> /tmp/rs.err
python - <<'PY'
import re, json
t = open('/tmp/rs.err').read()
prep = [float(x) for x in re.findall(r'prepare_problem total ([0-9.]+) ms', t)]
adj = [float(x) for x in re.findall(r'adjacency ([0-9.]+) ms', t)]
lists = [float(x) for x in re.findall(r'lists ([0-9.]+) ms', t)]
ana = re.findall(r'beside the (\w+) analysis \(([0-9.]+) ms\)', t)
d = json.loads(open('/tmp/rs.json').read().strip().splitlines()[-1])
print('closures', d['stats']['loop_closures'], 'solver_ms', d['ms_split']['solver'])
print('prepare_problem: n %d total %.1f ms mean %.3f' % (len(prep), sum(prep), sum(prep) / max(1, len(prep))))
print('adjacency total %.1f, lists total %.1f' % (sum(adj), sum(lists)))
for kind in ('incremental', 'full'):
    v = [float(b) for a, b in ana if a == kind]
    print(kind, 'analyses', len(v), 'total %.1f ms mean %.3f' % (sum(v), sum(v) / max(1, len(v))))
fr = re.findall(r'free nodes (\d+), fronts (\d+), levels (\d+)', t)
print('last problems:', fr[-3:])
PY
