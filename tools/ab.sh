#!/bin/bash
# A/B timing of engine switches on the headline workload: tools/ab.sh "name=0" "name=1" ...   (each arg = one bench run)
# prints ms/step (pipelined, timed region) and the serial per-class table
for o in "$@"; do
  extra=""
  for kv in $o; do [ "$kv" != "-" ] && extra="$extra --opt $kv"; done
  python bench.py --no-cpu-baseline --no-other-configs --no-pcie --steps 4 $extra 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][0])
r=d['roofline']; iso=r['roofline_isolated']
print('$o', 'ms/step %.2f' % d['ms_per_step'], 'serial %.2f' % iso['ms_per_step'])
print('  serial per class:', {k:v for k,v in iso['per_class_ms_per_step'].items()})
"
done
