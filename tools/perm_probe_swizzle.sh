#!/bin/bash
# Software XOR swizzle of the state's addresses: do the expensive index bits (17-23: rows of one DRAM bank) become cheap when
# they are also folded into the bank / channel bits (11-16)?  In place, gate-less, realistic and worst-case tile sets.
n=${1:-28}
R=$(cd "$(dirname "$0")/.." && pwd)
sets=("3,4,5,6,12,13,14,15" "3,4,5,6,20,21,22,23" "3,4,5,6,18,19,20,21" "4,6,7,12,14,18,21,25" "15,17,21,22,23,25,26,27" "7,8,10,18,19,20,22,23" "10,11,16,19,21,23,26,27" "4,5,6,13,14,15,18,20" "7,19,21,22,23,24,26,27" "8,9,17,19,23,24,25,27" "3,4,5,6,7,8,9,10" "4,7,8,9,10,11,14,16")
swz=("" "17>11,18>12,19>13,20>14,21>15,22>16" "18>12,19>13,20>14,21>15,22>16,23>17" "17>10,18>11,19>12,20>13,21>14,22>15,23>16" "20>12,21>13,22>14,23>15" "18>11,19>12,20>13,21>14,22>15,23>16,24>17,25>10" "17>3,18>4,19>5,20>6,21>7,22>8,23>9")
specs=()
for z in "${swz[@]}"; do
  for s in "${sets[@]}"; do
    zz=""; [ -n "$z" ] && zz=";swz=$z"
    specs+=("R=$s;inplace$zz;name=swz[${z:-none}] R=$s")
  done
done
"$R/tools/perm_probe" $n "${specs[@]}"
