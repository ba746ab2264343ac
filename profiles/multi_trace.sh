#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of the in-process sharded flow, R ranks sharing the one GPU (profiles/multi_rehearsal.py).
#   gpurun --timeout 900 -- 'bash profiles/multi_trace.sh r02_multi2 1.0 2'
set -e
set -o pipefail
TAG=${1:-r02_multi}
SCALE=${2:-1.0}
RANKS=${3:-2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- python3 $ROOT/profiles/multi_rehearsal.py $SCALE $RANKS only > $OUT/${TAG}_trace.log 2>&1
