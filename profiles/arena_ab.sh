set -e
for i in 1 2; do
  echo "== arena"; CORA_TRACE=1 RS_TRACE=1 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-extras 2> gpurun_out/ab_arena_$i.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms'], d['roofline']['achieved'])"
  echo "== no arena"; CORA_ARENA=0 RS_NO_PLACEMENT=1 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-extras 2> gpurun_out/ab_plain_$i.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms'], d['roofline']['achieved'])"
done
