#!/usr/bin/env bash
# One short headline bench (placed buffer, probe with escalation) in a FRESH process on whatever box the call lands on; run several times
# from the build container:  for i in 1 2 3 ...; do gpurun -- 'bash profiles/r05_bench_boxes.sh'; done  -> one line per box
python3 bench.py --no-cpu-baseline --no-extra --no-live-traffic 2>/dev/null | python3 -c "
import json, sys, socket
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
p = d['placement']['rollout']
print('box %s: %.2f us/step frac %.3f unplaced %.3f | kept %s %.4f ms, as created %.4f, stages %s, probe %.1f s, retired %.0f GB' % (
    socket.gethostname()[-6:], d['ms_per_step'] * 1e3, d['roofline']['frac'], d['roofline']['frac_unplaced'], p['kept'], p['kept_ms'], p['as_created_ms'],
    [(s.get('arena_GB'), s.get('kept_ms')) for s in p['stages']], p['probe_seconds'], p['retired_address_space_GB']))"
