#!/bin/bash
n=${1:-28}
R=$(cd "$(dirname "$0")/.." && pwd)
P="3,4,5,6,12,13,14,15"
specs=()
for D in "7,8,9,10,11,16,17,18" "8,11,16,20,21,25,26,27" "7,8,10,18,19,20,22,23" "16,17,18,19,20,21,22,23"; do
  swap=""
  IFS=',' read -ra PA <<< "$P"; IFS=',' read -ra DA <<< "$D"
  for i in 0 1 2 3 4 5 6 7; do swap="$swap${PA[$i]}>${DA[$i]},${DA[$i]}>${PA[$i]},"; done
  swap=${swap%,}
  for pl in 0 2 3 1; do
    specs+=("R=$D;inplace;plain=$pl;name=inplace D=$D plain=$pl")
    specs+=("R=$P;P=$swap;plain=$pl;name=v1 P*->D D=$D plain=$pl")
    specs+=("R=$D;P=$swap;plain=$pl;name=v2 D->P* D=$D plain=$pl")
  done
done
for pl in 0 2 3 1; do specs+=("R=$P;inplace;plain=$pl;name=inplace P* plain=$pl"); done
"$R/tools/perm_probe" $n "${specs[@]}"
