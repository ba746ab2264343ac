#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel trace + stats of the default bench, then the two PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, counter slots), each a run of its own.
#   gpurun --timeout 900 -- 'bash profiles/collect.sh r01_b'
# Raw output lands under gpurun_out/<tag>_*; profiles/summarize.py turns it into the committed summaries.
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- python3 $ROOT/bench.py --steps 5 --warmup 1 --cpu-sample 0 --collapse-reads 0 > $OUT/${TAG}_trace.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -o f -- python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --collapse-reads 0 > $OUT/${TAG}_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -o w -- python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --collapse-reads 0 > $OUT/${TAG}_write.log 2>&1
ls $OUT/${TAG}_trace $OUT/${TAG}_fetch $OUT/${TAG}_write
