#!/bin/bash
# builds the per-instance harness binaries scripts/bf3_round.sh runs (tools/bf3bench6.hip: a bf16x3 instance against the fp32-MFMA instance of
# the same layer).  Usage: scripts/build_bf3bench.sh [names...] (default: all)
cd "$(dirname "$0")/.."
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=12"
declare -A F
F[l7]="-DKINDV=0 -DLV=7 -DGSV=32 -DFMS=16 -DBMS=32"
F[l7r]="-DKINDV=0 -DLV=7 -DGSV=32 -DFMS=16 -DBMS=32 -DRESV=1"
F[l7s]="-DKINDV=0 -DLV=7 -DGSV=32 -DFMS=16 -DBMS=32 -DEDMP_BF3_STAMPS"
F[l7g16]="-DKINDV=0 -DLV=7 -DGSV=16 -DFMS=16 -DBMS=16"
F[l7g16r]="-DKINDV=0 -DLV=7 -DGSV=16 -DFMS=16 -DBMS=16 -DRESV=1"
F[l13]="-DKINDV=0 -DLV=13 -DGSV=16 -DFMS=16 -DBMS=16"
F[l13r]="-DKINDV=0 -DLV=13 -DGSV=16 -DFMS=16 -DBMS=16 -DRESV=1"
F[k4a]="-DKINDV=0 -DFKIND=4 -DLV=4 -DGSV=64 -DFMS=32 -DBMS=32 -DFCG=64 -DBCG=64"
F[k4ar]="-DKINDV=0 -DFKIND=4 -DLV=4 -DGSV=64 -DFMS=32 -DBMS=32 -DFCG=64 -DBCG=64 -DRESV=1"
F[k4b]="-DKINDV=0 -DFKIND=4 -DLV=4 -DGSV=32 -DFMS=32 -DBMS=32"
F[k4br]="-DKINDV=0 -DFKIND=4 -DLV=4 -DGSV=32 -DFMS=32 -DBMS=32 -DRESV=1"
F[d256]="-DKINDV=1 -DLV=7 -DGSV=32 -DFMS=32 -DBMS=32"
F[u256]="-DKINDV=2 -DLV=4 -DGSV=32 -DFMS=16 -DBMS=32"
F[d512]="-DKINDV=1 -DLV=4 -DGSV=64 -DFMS=32 -DBMS=32 -DFCG=64 -DBCG=64"
F[u512]="-DKINDV=2 -DLV=2 -DGSV=64 -DFMS=32 -DBMS=32 -DFCG=64 -DBCG=64"
F[d128]="-DKINDV=1 -DLV=13 -DGSV=16 -DFMS=16 -DBMS=16"
F[u128]="-DKINDV=2 -DLV=7 -DGSV=16 -DFMS=16 -DBMS=16"
NAMES=${@:-${!F[@]}}
for n in $NAMES; do
  ( $H ${F[$n]} tools/bf3bench6.hip -o tools/bf3bench6_$n > /tmp/bf3bench6_$n.log 2>&1 || { echo "FAILED $n"; grep -m3 error /tmp/bf3bench6_$n.log; } ) &
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 1; done
done
wait
ls tools/bf3bench6_* | grep -v "\.hip" | wc -l
