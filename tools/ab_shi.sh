#!/bin/bash
# A/B of the corner fixpoint schedule on ONE box.  usage: tools/ab_shi.sh "tiled,list,tail:inner ..." [reps]
CFGS=$1; REPS=${2:-2}
for rep in $(seq $REPS); do
  for cfg in $CFGS; do
    export SFMX_SHI_SWEEPS=${cfg%%:*} SFMX_SHI_INNER=${cfg##*:}
    echo "sweeps=$SFMX_SHI_SWEEPS inner=$SFMX_SHI_INNER: $(python bench.py --no-cpu-baseline --batched-probe 0 --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['host_seconds_per_step']; r=d['roofline']; print(d['value'], d['ms_per_step'], 'pf_busy', h['sec_pf_busy'], 'pf_gpu', h['sec_pf_gpu'], 'pf_replay', h['sec_pf_replay'], 'shi_wait', h['sec_shi_wait'], 'shi_us', round(r['kernel_us_per_pass'].get('shi fixpoint (k_shi_round / k_shi_list_* / k_shi_tail)',0)), d['passes_bit_identical'])")"
  done
done
