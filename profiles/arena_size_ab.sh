# the arena's size (one box, alternating), second word's chain of five arrays throughout
set -e
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms'], d['roofline']['achieved'])"; }
for g in 196 190 200 236; do
echo "== $g GiB"; CORA_B_CHAIN=5 CORA_TRACE=1 CORA_ARENA_GB=$g timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-extras 2> gpurun_out/as_$g.err | show; grep pool gpurun_out/as_$g.err | tail -1
done
