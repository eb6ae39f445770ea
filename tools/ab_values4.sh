#!/bin/bash
# usage: tools/ab_values4.sh VAR "v1 v2 .." [reps]  -- bench.py under several values of VAR; prints throughput, prefetch lane and replay times
VAR=$1; VALS=$2; REPS=${3:-2}
for rep in $(seq $REPS); do
  for v in $VALS; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    echo "$VAR=$v: $(python bench.py --no-cpu-baseline --batched-probe 0 --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['host_seconds_per_step']; r=d['roofline']; print(d['value'], d['ms_per_step'], 'lane_b', h['sec_lane_b_busy'], 'pf_busy', h['sec_pf_busy'], 'pf_gpu', h['sec_pf_gpu'], 'pf_replay', h['sec_pf_replay'], 'shi_wait', h['sec_shi_wait'], 'shi_us', round(r['kernel_us_per_pass'].get('shi fixpoint (k_shi_round / k_shi_list_* / k_shi_tail)',0)), d['passes_bit_identical'])")"
  done
done
