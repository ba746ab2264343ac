#!/bin/bash
# Runs ON THE GPU BOX: one rocprofv3 --pmc pass of SQ counters over one build of the bench workload (8 SQ counters fit one pass).
#   gpurun --timeout 600 -- 'bash profiles/pmc_sq.sh r02_sq [bench args]'
set -e
set -o pipefail
TAG=${1:-r02_sq}
shift || true
ARGS="$@"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/${TAG} -o s -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-extras $ARGS > $OUT/${TAG}.log 2>&1
