#!/bin/bash
# A/B on ONE box: bench.py under several values of one environment variable, alternating.
# usage: tools/ab_values.sh VAR "v1 v2 v3" [reps]     ("-" = unset)
VAR=$1; VALS=$2; REPS=${3:-2}
for rep in $(seq $REPS); do
  for v in $VALS; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    echo "$VAR=$v: $(python bench.py --no-cpu-baseline --batched-probe 0 --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['host_seconds_per_step']; print(d['value'], d['ms_per_step'], 'shi_wait', h['sec_shi_wait'], 'pf_busy', h['sec_pf_busy'], 'join', h['sec_join_wait'], 'feed_wait', h['sec_feed_wait'], 'identical', d['passes_bit_identical'])")"
  done
done
