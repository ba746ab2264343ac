# the arena's size (one box, one after the other): the default 80 B per base + 6 GB (236 GiB at 3.09 Gbp), smaller and larger ones.
# Below about 201 GiB the second word's chain of five arrays does not fit and the sort falls back to three or to a ping-pong
# (CORA_TRACE's pool line says how much was in use: 182 GB with five arrays, 136 with three, 113 with two).
set -e
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms'], d['roofline']['achieved'])"; }
for g in 236 190 200 225 250 236; do
echo "== $g GiB"; CORA_TRACE=1 CORA_ARENA_GB=$g timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-extras 2> gpurun_out/as_$g.err | show; grep pool gpurun_out/as_$g.err | tail -1
done
