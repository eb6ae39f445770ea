#!/bin/bash
# A/B on ONE box: bench.py under several values of one environment variable, alternating (boxes of the pool differ by +-5 %,
# and so do runs on one box: use >= 3 repetitions).  usage: tools/ab_values.sh VAR "v1 v2 v3" [reps]     ("-" = unset)
VAR=$1; VALS=$2; REPS=${3:-3}
for rep in $(seq $REPS); do
  for v in $VALS; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    echo "$VAR=$v: $(python bench.py --no-cpu-baseline --batched-probe 0 --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['host_seconds_per_step']; r=d['roofline']
shi=r['kernel_us_per_pass'].get('shi fixpoint (k_shi_round / k_shi_list_* / k_shi_tail)',0)
print(d['value'], d['ms_per_step'], 'lane_b', h['sec_lane_b_busy'], 'm_step', h['sec_m_step'], 'klt_lane', h['sec_klt'], 'pf_busy', h['sec_pf_busy'],
      'pf_replay', h['sec_pf_replay'], 'shi_wait', h['sec_shi_wait'], 'shi_us', round(shi), r['kernel'], r['avg_launch_us'], 'identical', d['passes_bit_identical'])")"
  done
done
