#!/bin/bash
# How the allocations are made and how fast the scatter runs through them (two-word records, one process per line):
# separate allocations of the plain sizes, sizes rounded to 2 MB, and all arrays cut from one allocation.
set -e
make -C cora_amd/csrc tools > /dev/null 2>&1
run() { RS_TRACE=1 timeout -k 10 100 cora_amd/bin/sortbench 2600000000 1 sorts 2 2>&1 | sed -e 's/(result.*//' | grep pass | awk '{print $3, $5}' | paste - - - -; }
for i in 1 2 3 4; do
  echo "== separate allocations"; run
  echo "== one allocation, arrays at multiples of 2 MB"; SB_ONE=1 SB_ROUND=2097152 run
  echo "== one allocation, arrays at multiples of 1 GB"; SB_ONE=1 SB_ROUND=1073741824 run
done
