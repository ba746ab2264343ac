#!/bin/bash
# Builds the real reference homTable_setup / homTable_setup_L from the sources
# where they lie under /root/reference into oracle/_ref/ (git-ignored, travels
# to the GPU box with the snapshot).  TEST INFRASTRUCTURE ONLY: used to make the
# golden fixtures, to validate the C restatement, and as bench.py's CPU baseline.
# Flags follow /root/reference/makefile:16-19 minus -march=native (the binary
# must also run on the GPU box's host CPU).
set -e
here="$(cd "$(dirname "$0")" && pwd)"
src="${CORA_REFERENCE:-/root/reference}/homTable_setup.cpp"
[ -f "$src" ] || { echo "reference source not present ($src): skipping _ref build"; exit 0; }
mkdir -p "$here/_ref"
g++ -static -O3 -w -o "$here/_ref/homTable_setup"   "$src" -fopenmp 2>/dev/null
g++ -DCHR_SHORT -static -O3 -w -o "$here/_ref/homTable_setup_L" "$src" -fopenmp 2>/dev/null
echo "built $here/_ref/homTable_setup and homTable_setup_L"
