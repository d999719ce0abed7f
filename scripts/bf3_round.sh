#!/bin/bash
# bf16x3 kernels (csrc/bf3.hip) vs the production fp32-MFMA instances, per instance, on one box (tools/bf3bench6.hip)
O=gpurun_out/${ROUND:-r06}
mkdir -p $O
{
for w in ${WK:-0 1 2 3}; do timeout 120 tools/bf3bench6_l7 256 $w; done
timeout 120 tools/bf3bench6_l7r 128 0
for w in ${WK:-0 1 2 3}; do timeout 120 tools/bf3bench6_l7g16 128 $w; done
timeout 120 tools/bf3bench6_l7g16r 512 0 1
for w in ${WK:-0 1 2 3}; do timeout 120 tools/bf3bench6_l13 128 $w; done
timeout 120 tools/bf3bench6_l13r 64 0
timeout 120 tools/bf3bench6_l7 256 0 0 1000
timeout 120 tools/bf3bench6_l13 128 0 0 37
timeout 120 tools/bf3bench6_l7s 256 0
} 2>&1 | tee $O/bf3bench6.txt
{
for x in d256 u256 d512 u512 d128 u128; do for w in ${WK:-0 1 2 3}; do timeout 120 tools/bf3bench6_$x $(case $x in *512) echo 512;; *256) echo 256;; *) echo 128;; esac) $w; done; done
} 2>&1 | tee $O/bf3bench6_resamplers.txt
{
for w in ${WK:-0 1 2 3}; do timeout 120 tools/bf3bench6_k4a 512 $w; done
timeout 120 tools/bf3bench6_k4ar 256 0
for w in ${WK:-0 1 2 3}; do timeout 120 tools/bf3bench6_k4b 256 $w; done
timeout 120 tools/bf3bench6_k4br 1024 0 1
} 2>&1 | tee $O/bf3bench6_l4.txt
