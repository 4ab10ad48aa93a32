#!/bin/bash
# usage: tools/sweep.sh "<chunk list>" [extra bench args]
for c in $1; do
  timeout 600 python bench.py --chunk $c --steps 2 --warmup 1 --no-cpu-baseline ${@:2} 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('chunk $c', round(d['value']), round(d['ms_per_step'],1))"
done
