#!/bin/bash
# Sliding windows of tile bits: which index bits are cheap for the WRITE side / READ side of a tile pass
n=${1:-28}
R=$(cd "$(dirname "$0")/.." && pwd)
specs=()
for mode in wo ro rw; do
  m=";$mode"; [ $mode = rw ] && m=""
  for x in 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24; do
    specs+=("R=3,4,5,6,$x,$((x+1)),$((x+2)),$((x+3));inplace$m;name=$mode 3-6 + $x..$((x+3))")
  done
  # the window bits as the ELEMENT bits of a thread (8 accesses of one thread) vs as wave bits: keep 12-15 fixed, vary low part
  for x in 3 4 5 6 7 8; do
    specs+=("R=$x,$((x+1)),$((x+2)),$((x+3)),12,13,14,15;inplace$m;name=$mode $x..$((x+3)) + 12-15")
  done
  for x in 16 18 20 22 24; do
    specs+=("R=3,4,5,6,12,13,$x,$((x+1));inplace$m;name=$mode 3-6,12,13 + $x,$((x+1))")
    specs+=("R=3,4,5,6,$x,$((x+1)),$((x+2)),$((x+3));inplace$m;order=2;name=$mode 3-6 + $x..$((x+3)) bit-reversed order")
  done
done
"$R/tools/perm_probe" $n "${specs[@]}"
