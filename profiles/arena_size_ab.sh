# the arena's size (one box, alternating): the default 80 B per base + 6 GB (236 GiB at 3.09 Gbp), 225 GiB, 250 GiB
set -e
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms'], d['roofline']['achieved'])"; }
for i in 1 2; do
echo "== default"; CORA_TRACE=1 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-extras 2> gpurun_out/as_def_$i.err | show
echo "== 225 GiB"; CORA_TRACE=1 CORA_ARENA_GB=225 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-extras 2> gpurun_out/as_225_$i.err | show
echo "== 250 GiB"; CORA_TRACE=1 CORA_ARENA_GB=250 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-extras 2> gpurun_out/as_250_$i.err | show
done
