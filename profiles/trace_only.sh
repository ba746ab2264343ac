#!/bin/bash
# Runs ON THE GPU BOX: kernel trace + stats of one bench configuration only (no PMC passes).
#   gpurun --timeout 900 -- 'bash profiles/trace_only.sh r02_b_3g [bench args]'
set -e
set -o pipefail
TAG=${1:-r02}
shift || true
ARGS="$@"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-extras $ARGS > $OUT/${TAG}_trace.log 2>&1
