#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel trace + stats of one collapse run (64 Mbp reference, 10 M reads of 100 bases).
#   gpurun --timeout 600 -- 'bash profiles/collect_collapse.sh r01_f'
# The profiler's per-kernel table is copied to profiles/<tag>_collapse_kernel_stats.csv by the caller.
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export CORA_CB_ONE=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_collapse_trace -o c -- python3 $ROOT/profiles/collapse_bench.py 10000000 64000000 100 > $OUT/${TAG}_collapse_trace.log 2>&1
find $OUT/${TAG}_collapse_trace -name '*kernel_stats.csv'
