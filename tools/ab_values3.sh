#!/bin/bash
# usage: tools/ab_values3.sh VAR "v1 v2 .." [reps]  -- bench.py under several values of VAR; prints throughput and the KLT launch time
VAR=$1; VALS=$2; REPS=${3:-2}
for rep in $(seq $REPS); do
  for v in $VALS; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    echo "$VAR=$v: $(python bench.py --no-cpu-baseline --batched-probe 0 --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['host_seconds_per_step']; r=d['roofline']; print(d['value'], d['ms_per_step'], 'lane_b', h['sec_lane_b_busy'], 'klt_lane', h['sec_klt'], 'm_step', h['sec_m_step'], 'roofline', r['kernel'], r['avg_launch_us'], r['frac'])")"
  done
done
