"""Print ms_per_step of a bench.py JSON line read from stdin (last line)."""
import json, sys
line = [l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]
d = json.loads(line)
print('%.2f ms/step  %s' % (d['ms_per_step'], d['config'].get('workload')))
