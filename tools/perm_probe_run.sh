#!/bin/bash
# Sweep of tools/perm_probe (round 3): which read layouts of a tile are fast, what an out-of-place pass costs, and
# what a pass costs whose store layout is a bit permutation of its load layout.   bash tools/perm_probe_run.sh [n]
n=${1:-28}
R=$(cd "$(dirname "$0")/.." && pwd)
P="3,4,5,6,12,13,14,15"
specs=(
 # in place, read layout = write layout (a gate-less k_tile pass)
 "R=3,4,5,6,7,8,9,10;inplace;name=inplace contiguous 3-10"
 "R=$P;inplace;name=inplace P*=3-6,12-15"
 "R=3,4,5,6,20,21,22,23;inplace;name=inplace 3-6,20-23"
 "R=3,4,5,6,16,17,18,19;inplace;name=inplace 3-6,16-19"
 "R=4,6,7,12,14,18,21,25;inplace;name=inplace bench pass 4"
 "R=4,5,6,13,14,15,18,20;inplace;name=inplace bench pass 13"
 "R=15,17,21,22,23,25,26,27;inplace;name=inplace bench pass 2"
 "R=4,7,8,9,10,11,14,16;inplace;name=inplace bench pass 0"
 # where is the sweet spot of the read layout
 "R=3,4,5,6,11,12,13,14;inplace;name=inplace 3-6,11-14"
 "R=3,4,5,6,13,14,15,16;inplace;name=inplace 3-6,13-16"
 "R=3,4,5,6,14,15,16,17;inplace;name=inplace 3-6,14-17"
 "R=3,4,5,6,7,12,13,14;inplace;name=inplace 3-7,12-14"
 "R=3,4,5,12,13,14,15,16;inplace;name=inplace 3-5,12-16"
 "R=3,4,12,13,14,15,16,17;inplace;name=inplace 3-4,12-17"
 "R=12,13,14,15,16,17,18,19;inplace;name=inplace 12-19"
 "R=3,4,5,6,7,8,12,13;inplace;name=inplace 3-8,12-13"
 "R=3,4,5,6,9,10,11,12;inplace;name=inplace 3-6,9-12"
 "R=3,4,5,6,10,11,12,13;inplace;name=inplace 3-6,10-13"
 # reads only / writes only
 "R=$P;inplace;ro;name=reads only P*"
 "R=$P;inplace;wo;name=writes only P*"
 "R=3,4,5,6,7,8,9,10;inplace;ro;name=reads only contiguous"
 "R=3,4,5,6,7,8,9,10;inplace;wo;name=writes only contiguous"
 "R=3,4,5,6,20,21,22,23;inplace;ro;name=reads only 3-6,20-23"
 "R=3,4,5,6,20,21,22,23;inplace;wo;name=writes only 3-6,20-23"
 "R=4,5,6,13,14,15,18,20;inplace;ro;name=reads only bench pass 13"
 "R=4,5,6,13,14,15,18,20;inplace;wo;name=writes only bench pass 13"
 # out of place, same layout
 "R=$P;name=out-of-place identity P*"
 "R=3,4,5,6,7,8,9,10;name=out-of-place identity contiguous"
 "R=3,4,5,6,20,21,22,23;name=out-of-place identity 3-6,20-23"
 # out of place with a permutation: read P*, the next tile's four new bits come from H, four old ones leave
 "R=$P;P=12>20,13>21,14>22,15>23,20>12,21>13,22>14,23>15;name=perm swap 12-15<->20-23"
 "R=$P;P=12>7,13>8,14>9,15>10,7>20,8>21,9>22,10>23,20>12,21>13,22>14,23>15;name=perm cycle 12-15>7-10>20-23>12-15"
 "R=$P;P=12>16,13>17,14>18,15>19,16>12,17>13,18>14,19>15;name=perm swap 12-15<->16-19"
 "R=$P;P=12>7,13>8,14>9,15>10,7>16,8>17,9>18,10>19,16>12,17>13,18>14,19>15;name=perm cycle 12-15>7-10>16-19>12-15"
 "R=$P;P=12>24,13>25,14>26,15>27,24>12,25>13,26>14,27>15;name=perm swap 12-15<->24-27"
 "R=$P;P=12>7,13>8,14>9,15>10,7>24,8>25,9>26,10>27,24>12,25>13,26>14,27>15;name=perm cycle 12-15>7-10>24-27>12-15"
 "R=$P;P=12>9,13>17,14>21,15>26,9>12,17>13,21>14,26>15;name=perm swap 12-15<->9,17,21,26"
 # all eight tile bits replaced
 "R=$P;P=3>20,4>21,5>22,6>23,12>24,13>25,14>26,15>27,20>3,21>4,22>5,23>6,24>12,25>13,26>14,27>15;name=perm swap all 8 <-> 20-27"
 "R=$P;P=3>7,4>8,5>9,6>10,12>11,13>16,14>17,15>18,7>20,8>21,9>22,10>23,11>24,16>25,17>26,18>27,20>3,21>4,22>5,23>6,24>12,25>13,26>14,27>15;name=perm cycle all 8 via 7-11,16-18"
 # the opposite scheme: scattered reads (any layout), stores always in the P* layout
 "R=3,4,5,6,20,21,22,23;P=20>12,21>13,22>14,23>15,12>20,13>21,14>22,15>23;name=gather 3-6,20-23 -> P*"
 "R=4,5,6,13,14,15,18,20;P=13>12,14>13,15>14,18>15,20>3,3>18,12>20;name=gather bench pass 13 -> 3-6,12-15"
 # tile order / tiles per workgroup on the permuted pass
 "R=$P;P=12>7,13>8,14>9,15>10,7>20,8>21,9>22,10>23,20>12,21>13,22>14,23>15;order=1;name=perm cycle 20-23, XCD-contiguous"
 "R=$P;P=12>7,13>8,14>9,15>10,7>20,8>21,9>22,10>23,20>12,21>13,22>14,23>15;tpw=1;name=perm cycle 20-23, tpw=1"
 "R=$P;inplace;order=1;name=inplace P* XCD-contiguous"
 "R=$P;inplace;lds=0;name=inplace P* no LDS"
)
"$R/tools/perm_probe" $n "${specs[@]}"
