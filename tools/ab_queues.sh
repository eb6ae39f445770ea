#!/bin/bash
# A/B of (lane creation order, hardware queue count) on ONE box.  usage: tools/ab_queues.sh "ORDER:Q ORDER:Q ..." [reps]
CFGS=$1; REPS=${2:-2}
export SFMX_PREFETCH_WORKERS=${SFMX_PREFETCH_WORKERS:-2}
for rep in $(seq $REPS); do
  for cfg in $CFGS; do
    export SFMX_LANE_ORDER=${cfg%%:*} GPU_MAX_HW_QUEUES=${cfg##*:}
    echo "order=$SFMX_LANE_ORDER queues=$GPU_MAX_HW_QUEUES: $(python bench.py --no-cpu-baseline --batched-probe 0 --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['host_seconds_per_step']; print(d['value'], d['ms_per_step'], 'lane_b', h['sec_lane_b_busy'], 'pf', h['sec_pf_busy'], 'klt', h['sec_klt'], 'm_step', h['sec_m_step'], 'feed_wait', h['sec_feed_wait'])")"
  done
done
