# Runs ON THE GPU BOX: seeds stage with four global passes (CORA_SEED_LDS=0) against three + the LDS finish (2) at three sizes of the
# bench generator, then the earlier-seed test of the sketch filter on and off at full size.
set -e
cd $GRAFT_REPO_ROOT
for sc in 0.17 0.33 0.5; do
 for m in 0 2; do
  echo "scale $sc CORA_SEED_LDS=$m"
  CORA_SEED_LDS=$m timeout -k 10 200 python -u bench.py --scale $sc --steps 3 --warmup 1 --no-extras 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['ms_per_step'], j['stage_ms']['ms_seeds'], j['config']['representatives_rank0'])"
 done
done
for e in 1 0; do
 echo "scale 1 CORA_SK_EARLY=$e"
 CORA_SK_EARLY=$e timeout -k 10 300 python -u bench.py --steps 4 --warmup 1 --no-extras 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['ms_per_step'], j['stage_ms'], j['config']['hom_pairs'])"
done
