#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel trace + stats of the default bench, then the two PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, counter slots), each a run of its own.
#   gpurun --timeout 900 -- 'bash profiles/collect.sh r02_a_3g'                       (the default bench workload: 3.09 Gbp)
#   gpurun --timeout 900 -- 'bash profiles/collect.sh r02_a_64m --workload chr20_like'  (further arguments go to bench.py)
# Raw output lands under gpurun_out/<tag>_*; profiles/summarize.py turns it into the committed summaries.
set -e
set -o pipefail
TAG=${1:-r02}
shift || true
ARGS="$@"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-extras $ARGS > $OUT/${TAG}_trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -o f -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-extras $ARGS > $OUT/${TAG}_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -o w -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-extras $ARGS > $OUT/${TAG}_write.log 2>&1
ls $OUT/${TAG}_trace $OUT/${TAG}_fetch $OUT/${TAG}_write
