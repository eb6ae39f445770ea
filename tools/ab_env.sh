#!/bin/bash
# A/B on ONE box: bench.py with and without an environment switch, alternating.  usage: tools/ab_env.sh VAR [reps]
VAR=$1; REPS=${2:-2}
for rep in $(seq $REPS); do
  for on in 0 1; do
    if [ $on = 1 ]; then export $VAR=1; else unset $VAR; fi
    echo "$VAR=$on: $(python bench.py --no-cpu-baseline --batched-probe 0 --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['host_seconds_per_step']; print(d['value'], d['ms_per_step'], 'lane_b', h['sec_lane_b_busy'], 'join', h['sec_join_wait'], 'identical', d['passes_bit_identical'])")"
  done
done
