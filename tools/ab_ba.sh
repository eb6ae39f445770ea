#!/bin/bash
# A/B of the BA launch shapes on ONE box, alternating: fused reduce+solve on/off x expansion merged/split
for rep in 1 2; do
  for cfg in "merged:" "split:" "merged:1" "split:1"; do
    export SFMX_BA_EXPAND=${cfg%%:*}
    if [ -n "${cfg##*:}" ]; then export SFMX_BA_NO_FUSE=1; else unset SFMX_BA_NO_FUSE; fi
    echo "expand=$SFMX_BA_EXPAND no_fuse=${SFMX_BA_NO_FUSE:-0}: $(SFMX_PREFETCH_WORKERS=2 python bench.py --no-cpu-baseline --batched-probe 0 --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['host_seconds_per_step']; print(d['value'], d['ms_per_step'], 'lane_b', h['sec_lane_b_busy'], 'join', h['sec_join_wait'], 'm_step', h['sec_m_step'], 'identical', d['passes_bit_identical'])")"
  done
done
