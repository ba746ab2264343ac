# which end of the arena the big arrays are cut from (one box, alternating): the default (first sort from the high end, the rest
# from the low end), the low end throughout, the high end throughout
set -e
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms'], d['roofline']['achieved'])"; }
for i in 1 2; do
echo "== default"; timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-extras 2> /dev/null | show
echo "== low end"; CORA_ARENA_HIGH=0 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-extras 2> /dev/null | show
echo "== high end"; CORA_ARENA_HIGH=1 RS_TRACE=1 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-extras 2> gpurun_out/end_high_$i.err | show
done
