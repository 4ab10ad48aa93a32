#!/bin/bash
# usage: tools/sweep2.sh "<arg sets separated by ;>"   e.g. tools/sweep2.sh "--no-fuse;;--overlap 0"
IFS=';' read -ra SETS <<< "$1"
for a in "${SETS[@]}"; do
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $a 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$a]', round(d['value']), round(d['ms_per_step'],1), {k:v for k,v in d['roofline']['per_class_ms_per_step'].items() if k.startswith('gru64') or 'fc_ln' in k})"
done
