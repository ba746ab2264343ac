set -e
cd $GRAFT_REPO_ROOT
for e in 1 0 1 0; do
 echo "scale 1 CORA_SK_EARLY=$e"
 CORA_SK_EARLY=$e timeout -k 10 300 python -u bench.py --steps 4 --warmup 1 --no-extras 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['ms_per_step'], j['stage_ms'], j['config']['hom_pairs'])"
done
