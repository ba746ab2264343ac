# bench.py with the stage arrays cut from one allocation (DevPool's arena, the default) and from separate blocks, alternating
# on one box.  CORA_TRACE / RS_TRACE: pool figures and every pass's scatter time with its arrays' addresses.
set -e
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms'], d['roofline']['achieved'])"; }
for i in 1 2; do
  echo "== arena"; CORA_TRACE=1 RS_TRACE=1 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-extras 2> gpurun_out/ab_arena_$i.err | show
  echo "== separate blocks"; CORA_ARENA=0 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-extras 2> gpurun_out/ab_plain_$i.err | show
done
