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
# The consumer's table loaders (mappingInference.cpp:554-723, 1084-1153) behind a dump harness of our own
# (ref_loader_main.cpp includes the reference source from where it lies).  _L = 2-byte contig ids, _e3 = room for
# three offsets per item (-DED_THREE, mappingInference.cpp:20-23; the reference's makefile builds the same variants).
msrc="${CORA_REFERENCE:-/root/reference}/mappingInference.cpp"
if [ -f "$msrc" ]; then
  for v in "" "_L" "_e3" "_L_e3"; do
    fl=""; case "$v" in *_L*) fl="$fl -DCHR_SHORT";; esac; case "$v" in *_e3) fl="$fl -DED_THREE";; esac
    g++ -O2 -w $fl -DCORA_REF_SRC="\"$msrc\"" -I"$here" -o "$here/_ref/ref_loader$v" "$here/ref_loader_main.cpp" &
  done
  wait
  echo "built $here/_ref/ref_loader{,_L,_e3,_L_e3}"
fi
# collapse / collapse_L (read-side read-mer dedup): /root/reference/makefile:6-9 minus -march=native
csrc="${CORA_REFERENCE:-/root/reference}/collapse.cpp"
if [ -f "$csrc" ]; then
  g++ -static -O3 -w -o "$here/_ref/collapse" "$csrc" 2>/dev/null &
  g++ -DCHR_SHORT -static -O3 -w -o "$here/_ref/collapse_L" "$csrc" 2>/dev/null &
  wait
  echo "built $here/_ref/collapse and collapse_L"
fi
# fastqSplit (signal-keyed partitioning of the reads): /root/reference/makefile:12-13 minus -march=native
fsrc="${CORA_REFERENCE:-/root/reference}/fastqSplit.cpp"
if [ -f "$fsrc" ]; then
  g++ -static -O3 -w -o "$here/_ref/fastqSplit" "$fsrc" 2>/dev/null
  echo "built $here/_ref/fastqSplit"
fi
