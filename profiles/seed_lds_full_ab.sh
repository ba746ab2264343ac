# Runs ON THE GPU BOX: the 3.09 Gbp build with four-pass seed sorts (CORA_SEED_LDS=0) against the library's choice (three passes + LDS), twice each.
set -e
cd $GRAFT_REPO_ROOT
for v in "CORA_SEED_LDS=0" "CORA_SEED_LDS=1" "CORA_SEED_LDS=0" "CORA_SEED_LDS=1"; do
 echo "== $v"
 env $v timeout -k 10 300 python -u bench.py --steps 4 --warmup 1 --no-extras 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['ms_per_step'], j['stage_ms'], j['roofline']['avg_launch_ms'], j['roofline']['launches_per_step'])"
done
