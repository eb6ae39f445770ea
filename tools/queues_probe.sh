for q in 8 16 32; do echo "GPU_MAX_HW_QUEUES=$q: $(GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --batched-probe 3 --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('single', d['value'], 'batched3', d['batched']['value'])")"; done
echo "two processes at once (8 queues each):"
(python bench.py --no-cpu-baseline --batched-probe 0 --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('procA', d['value'])") & 
(python bench.py --no-cpu-baseline --batched-probe 0 --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('procB', d['value'])") &
wait
