# Runs ON THE GPU BOX: histogram passes that read the key words (RS_NEXT_DIGIT=0) against digit bytes left by the scatter before (1), 3.09 Gbp build.
set -e
cd $GRAFT_REPO_ROOT
for v in "RS_NEXT_DIGIT=0" "RS_NEXT_DIGIT=1" "RS_NEXT_DIGIT=0" "RS_NEXT_DIGIT=1"; do
 echo "== $v"
 env $v timeout -k 10 300 python -u bench.py --steps 4 --warmup 1 --no-extras 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['ms_per_step'], j['stage_ms'], j['roofline']['avg_launch_ms'])"
done
