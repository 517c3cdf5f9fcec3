#!/bin/bash
# Occupancy of the tile mover: does the read side (4.6 TB/s at four workgroups per CU) follow the workgroups per CU?
n=${1:-28}
R=$(cd "$(dirname "$0")/.." && pwd)
P="R=3,4,5,6,12,13,14,15;inplace"
specs=()
for mode in ro wo rw; do
  m=";$mode"; [ $mode = rw ] && m=""
  specs+=("$P$m;tpw=1;lds=0;pad=65536;name=$mode tpw1 noLDS 2 wg/CU")
  specs+=("$P$m;tpw=1;lds=0;pad=32768;name=$mode tpw1 noLDS 4 wg/CU")
  specs+=("$P$m;tpw=1;lds=0;pad=25600;name=$mode tpw1 noLDS 6 wg/CU")
  specs+=("$P$m;tpw=1;lds=0;pad=0;name=$mode tpw1 noLDS 8 wg/CU")
  specs+=("$P$m;tpw=2;lds=0;occ=4;pad=32768;name=$mode tpw2 noLDS 4 wg/CU")
  specs+=("$P$m;tpw=2;lds=0;occ=8;name=$mode tpw2 noLDS 8 wg/CU (64 VGPRs)")
  specs+=("$P$m;tpw=1;lds=1;name=$mode tpw1 LDS32K 4 wg/CU")
  specs+=("$P$m;tpw=2;lds=1;name=$mode tpw2 LDS32K 4 wg/CU (the product shape)")
  specs+=("$P$m;tpw=1;lds=2;occ=8;name=$mode tpw1 LDS16K split 8 wg/CU")
  specs+=("$P$m;tpw=2;lds=2;occ=8;name=$mode tpw2 LDS16K split 8 wg/CU")
  specs+=("$P$m;tpw=1;lds=2;occ=8;pad=16384;name=$mode tpw1 LDS16K split 4 wg/CU")
done
# the same for a slow in-place set and for the contiguous tile
for R8 in "3,4,5,6,20,21,22,23" "3,4,5,6,7,8,9,10" "4,6,7,12,14,18,21,25"; do
  specs+=("R=$R8;inplace;tpw=2;lds=1;name=rw $R8 product shape")
  specs+=("R=$R8;inplace;tpw=1;lds=2;occ=8;name=rw $R8 tpw1 LDS16K split 8 wg/CU")
  specs+=("R=$R8;inplace;tpw=1;lds=0;pad=0;name=rw $R8 tpw1 noLDS 8 wg/CU")
done
"$R/tools/perm_probe" $n "${specs[@]}"
