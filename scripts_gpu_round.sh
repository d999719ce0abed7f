#!/bin/bash
# run on the GPU box by gpurun: bench, rocprofv3 kernel trace and PMC passes of the same command (outputs -> gpurun_out/)
mkdir -p gpurun_out
python bench.py --steps 3 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -2 gpurun_out/bench.err | grep -v amdgpu.ids
cat gpurun_out/bench.json
export TMPDIR=/tmp
REPO=$PWD
CMD="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -o trace -- $CMD > $REPO/gpurun_out/prof_bench.json 2> $REPO/gpurun_out/prof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/pmc_fetch -o f -- $CMD > /dev/null 2> $REPO/gpurun_out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/pmc_write -o w -- $CMD > /dev/null 2> $REPO/gpurun_out/pmc_write.err
cd $REPO
ls gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write
python scripts/summarize_profiles.py gpurun_out
