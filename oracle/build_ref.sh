#!/bin/bash
# Builds the real reference programs from the sources where they lie under /root/reference into oracle/_ref/
# (git-ignored, travels to the GPU box with the snapshot).  TEST INFRASTRUCTURE ONLY: used to make the golden
# fixtures, to validate the C restatements, as bench.py's CPU baseline, and -- cora itself -- to run the unmodified
# driver over the drop-in executables (tests/test_gpu_cora_driver.py).
# Flags follow /root/reference/makefile minus -march=native (the binaries must also run on the GPU box's host CPU).
# The reference is licensed for academic, non-commercial use (its LICENSE): nothing of it is stored in this repository.
# A failed compile stops the script: its messages are kept in a log that is printed then.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
ref="${CORA_REFERENCE:-/root/reference}"
src="$ref/homTable_setup.cpp"
[ -f "$src" ] || { echo "reference source not present ($src): skipping _ref build"; exit 0; }
mkdir -p "$here/_ref"
log="$here/_ref/build.log"
: > "$log"
pids=()
outs=()
compile() {          # compile <output> <g++ arguments...>   (in the background; checked by finish)
  local out="$1"; shift
  rm -f "$out"
  ( g++ "$@" -o "$out" >> "$log" 2>&1 ) &
  pids+=($!)
  outs+=("$out")
}
finish() {           # waits for every compile started so far; any failure or missing output ends the script
  local i
  for i in "${!pids[@]}"; do
    if ! wait "${pids[$i]}" || [ ! -x "${outs[$i]}" ]; then
      echo "build_ref.sh: compiling ${outs[$i]} FAILED; compiler output:" >&2
      cat "$log" >&2
      exit 1
    fi
  done
  pids=(); outs=()
}
# homTable_setup / homTable_setup_L: makefile:16-19
compile "$here/_ref/homTable_setup"   -static -O3 -w "$src" -fopenmp
compile "$here/_ref/homTable_setup_L" -DCHR_SHORT -static -O3 -w "$src" -fopenmp
finish
echo "built $here/_ref/homTable_setup and homTable_setup_L"
# The consumer's table loaders (mappingInference.cpp:554-723, 1084-1153) behind a dump harness of our own
# (ref_loader_main.cpp includes the reference source from where it lies).  _L = 2-byte contig ids, _e3 = room for
# three offsets per item (-DED_THREE, mappingInference.cpp:20-23; the reference's makefile builds the same variants).
msrc="$ref/mappingInference.cpp"
if [ -f "$msrc" ]; then
  for v in "" "_L" "_e3" "_L_e3"; do
    fl=""; case "$v" in *_L*) fl="$fl -DCHR_SHORT";; esac; case "$v" in *_e3) fl="$fl -DED_THREE";; esac
    compile "$here/_ref/ref_loader$v" -O2 -w $fl -DCORA_REF_SRC="\"$msrc\"" -I"$here" "$here/ref_loader_main.cpp"
  done
  finish
  echo "built $here/_ref/ref_loader{,_L,_e3,_L_e3}"
fi
# collapse / collapse_L (read-side read-mer dedup): makefile:6-9
csrc="$ref/collapse.cpp"
if [ -f "$csrc" ]; then
  compile "$here/_ref/collapse"   -static -O3 -w "$csrc"
  compile "$here/_ref/collapse_L" -DCHR_SHORT -static -O3 -w "$csrc"
  finish
  echo "built $here/_ref/collapse and collapse_L"
fi
# fastqSplit (signal-keyed partitioning of the reads): makefile:12-13
fsrc="$ref/fastqSplit.cpp"
if [ -f "$fsrc" ]; then
  compile "$here/_ref/fastqSplit" -static -O3 -w "$fsrc"
  finish
  echo "built $here/_ref/fastqSplit"
fi
# cora, the driver (makefile:10-11): run UNMODIFIED over the drop-in executables by tests/test_gpu_cora_driver.py
dsrc="$ref/cora.cpp"
if [ -f "$dsrc" ]; then
  compile "$here/_ref/cora" -static -O3 -w "$dsrc"
  finish
  echo "built $here/_ref/cora"
fi
