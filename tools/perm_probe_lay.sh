#!/bin/bash
# Which tile bits a wave instruction / a thread's 8 accesses / the 4 waves cover: does the issue shape change the cost of slow tile sets?
n=${1:-28}
R=$(cd "$(dirname "$0")/.." && pwd)
specs=()
for R8 in "4,6,7,12,14,18,21,25" "3,4,5,6,20,21,22,23" "7,8,10,18,19,20,22,23" "15,17,21,22,23,25,26,27" "3,4,5,6,12,13,14,15"; do
  for lay in "0,1,2,3,4,5,6,7" "0,1,2,5,6,3,4,7" "0,1,2,6,7,3,4,5" "5,6,7,3,4,0,1,2" "0,1,5,2,6,3,4,7" "0,1,2,3,7,4,5,6"; do
    for mode in rw wo; do
      m=";$mode"; [ $mode = rw ] && m=""
      specs+=("R=$R8;inplace$m;lay=$lay;name=$mode R=$R8 lay=$lay")
    done
  done
done
"$R/tools/perm_probe" $n "${specs[@]}"
