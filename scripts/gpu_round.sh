#!/bin/bash
# Run on the GPU box by gpurun: bench.py, the rocprofv3 kernel trace of the same command, and the PMC passes
# (FETCH_SIZE / WRITE_SIZE / MFMA busy, each its own run, --kernel-trace only).  Outputs -> gpurun_out/\$ROUND/ (default r06); raw
# output stays there untouched, scripts/collect_round.sh copies the summaries into profiles/ and derives profiles/<round>_numbers.md.
O=gpurun_out/${ROUND:-r06}
mkdir -p $O
python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
python scripts/show_bench.py $O/bench.json | cut -c1-220 | head -4
# one bench line per single-GPU BASELINE config: C2 = guides [1,2,3]; C5's per-rank workload = the 8-guide ensemble as
# one logical batch in an RCCL world of one (the per-guided-step all-reduce really issued)
python bench.py --guides 1,2,3 --steps 3 --warmup 1 --no-cpu-baseline --no-two-scenes --no-problem-set > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --guides 1,2,3,4,5,10,11,13 --logical-batch --steps 3 --warmup 1 --no-cpu-baseline --no-two-scenes --no-problem-set > $O/bench_c5_n1.json 2> $O/bench_c5_n1.err
python bench.py --guides 1,2,3,4,5,10,11,13 --logical-batch --hook python --steps 3 --warmup 1 --no-cpu-baseline --no-two-scenes --no-problem-set --no-roofline --no-native-leg > $O/bench_c5_n1_pyhook.json 2> $O/bench_c5_n1_pyhook.err
export TMPDIR=/tmp
REPO=$PWD
CMD="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-two-scenes --no-problem-set --no-native-leg"  # (two scenes in flight would overlap kernels of two contexts: durations in the trace would no longer be one chain's)
cd /tmp
rm -rf $REPO/$O/prof $REPO/$O/pmc_*
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof -o trace -- $CMD > $REPO/$O/prof_bench.json 2> $REPO/$O/prof.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $REPO/$O/pmc_$c -o p -- python $REPO/scripts/pmc_unet_forward.py > /dev/null 2> $REPO/$O/pmc_$c.err
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $REPO/$O/pmc_mfma -o p -- python $REPO/scripts/pmc_unet_forward.py > /dev/null 2> $REPO/$O/pmc_mfma.err
cd $REPO
cp $(find $O/prof -name "trace_kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
python scripts/summarize_profiles.py $O | head -8
python scripts/summarize_pmc.py $O $O/pmc_hbm_traffic.json 3
python scripts/summarize_mfma_pmc.py $(find $O/pmc_mfma -name "*counter_collection.csv" | head -1) $O/pmc_mfma_util.md | tail -12
rm -f $(find $O -name "*kernel_trace.csv") $(find $O -name "*counter_collection.csv")
