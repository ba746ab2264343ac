# the arena's size (one box, alternating): the default 58 B per base + 6 GB (173 GB at 3.09 Gbp), 128 GB, 240 GB
set -e
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms'], d['roofline']['achieved'])"; }
for i in 1 2; do
echo "== default"; CORA_TRACE=1 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-extras 2> gpurun_out/as_def_$i.err | show
echo "== 128 GB"; CORA_TRACE=1 CORA_ARENA_GB=128 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-extras 2> gpurun_out/as_128_$i.err | show
echo "== 240 GB"; CORA_TRACE=1 CORA_ARENA_GB=240 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-extras 2> gpurun_out/as_240_$i.err | show
done
