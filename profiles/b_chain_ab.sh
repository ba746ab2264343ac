# the sketches' second word ping-pongs (1) or climbs through a chain of 3 / 5 arrays during the seed sorts (one box, alternating)
set -e
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms'], d['roofline']['achieved'], d['tables_of_last_step'] if 'tables_of_last_step' in d else '')"; }
for i in 1 2; do
for c in 1 3 5; do
echo "== chain of $c"; CORA_B_CHAIN=$c CORA_ARENA_GB=236 CORA_TRACE=1 RS_TRACE=1 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-extras 2> gpurun_out/bc_${c}_$i.err | show
done
done
