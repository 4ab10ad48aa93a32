# Headline step time against the chunk length (dpdf_set_chunk_frames; 0 = the automatic schedule).  usage on the GPU box: bash tools/chunk_sweep.sh "0 128 160 192 224 256" [extra bench.py flags]
for c in $1; do
  for rep in 1 2; do
    python bench.py --steps 5 --warmup 2 --no-isolated --no-other-configs --no-cpu-baseline --no-pcie --no-dist-selftest --no-parity --profile-steps 0 --chunk $c $2 2>/dev/null \
      | python -c "import sys, json; d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('chunk', $c, 'ms/step', round(d['ms_per_step'], 2))"
  done
done
