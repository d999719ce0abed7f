#!/bin/bash
# kernel-time table of 3 UNet forwards (B=1024) + the UNet parity tests
export TMPDIR=/tmp
REPO=$PWD
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet or fused or teacher_forced_steps or determinism" 2>&1 | tail -2
cd /tmp
d=$REPO/gpurun_out/ktime
rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $REPO/scripts/pmc_unet_forward.py > $REPO/gpurun_out/ktime.out 2>&1
python - <<PY
import csv,glob
f=glob.glob("$d/**/p_kernel_stats.csv",recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    n=r['Name'].split('(')[0].replace('edmp::','').replace('void ','')
    tot+=float(r['TotalDurationNs'])
    print(f"{n:48s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1000:8.1f} total_us={float(r['TotalDurationNs'])/1000:9.1f}")
print("TOTAL us per forward (4 forwards incl. warm-up):", tot/1000/4)
PY
cd $REPO
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
