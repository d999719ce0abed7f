#!/bin/bash
# helper run on the GPU box by gpurun: bench + rocprofv3 kernel trace (outputs under gpurun_out/)
set -x
mkdir -p gpurun_out
python bench.py --steps 3 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -3 gpurun_out/bench.err
cat gpurun_out/bench.json
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -o r1 -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $REPO/gpurun_out/prof_bench.json 2> $REPO/gpurun_out/prof.err
cd $REPO
tail -3 gpurun_out/prof.err
ls -R gpurun_out/prof | head -20
