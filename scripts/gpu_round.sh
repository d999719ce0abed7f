#!/bin/bash
# run on the GPU box by gpurun: bench, rocprofv3 kernel trace of the same command, PMC passes (outputs -> gpurun_out/)
mkdir -p gpurun_out
python bench.py --steps 3 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -2 gpurun_out/bench.err | grep -v amdgpu.ids
cat gpurun_out/bench.json
export TMPDIR=/tmp
REPO=$PWD
CMD="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline"
cd /tmp
rm -rf $REPO/gpurun_out/prof $REPO/gpurun_out/pmc_FETCH_SIZE $REPO/gpurun_out/pmc_WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -o trace -- $CMD > $REPO/gpurun_out/prof_bench.json 2> $REPO/gpurun_out/prof.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $REPO/gpurun_out/pmc_$c -o p -- python $REPO/scripts/pmc_unet_forward.py > /dev/null 2> $REPO/gpurun_out/pmc_$c.err
done
cd $REPO
rm -f gpurun_out/prof/trace_kernel_trace.csv gpurun_out/pmc_*/p_kernel_trace.csv
python scripts/summarize_profiles.py gpurun_out | head -12
python scripts/summarize_pmc.py gpurun_out gpurun_out/pmc_hbm_traffic.json 3
