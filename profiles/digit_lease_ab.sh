# Runs ON THE GPU BOX: where the sorts' digit bytes live -- bench step, K = 64 d = 3, and the executable end to end.
set -e
cd $GRAFT_REPO_ROOT
for i in 1 2; do
 timeout -k 10 300 python -u bench.py --steps 4 --warmup 1 --no-extras 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['ms_per_step'], j['stage_ms'])"
done
timeout -k 10 300 python -u profiles/config3_run.py 2>&1 | tail -n 1
E2E_TINY_FIRST=1 timeout -k 10 300 python -u profiles/e2e_variants.py 1.0 "" "" 2>&1 | grep "wall\|e2e_ms\|destroyed"
