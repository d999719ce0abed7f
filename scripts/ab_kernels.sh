#!/bin/bash
# A/B kernel timing on ONE box: every ${EDMP_AB_DIR:-scratch/ab}/*.so is copied over the library in turn (twice, interleaved)
export TMPDIR=/tmp
REPO=$PWD
cp edmp_amd/libedmp_hip.so /tmp/orig.so
for rep in 1 2; do
for lib in ${EDMP_AB_DIR:-scratch/ab}/*.so; do
  cp $lib edmp_amd/libedmp_hip.so
  d=/tmp/ab_out; rm -rf $d
  (cd /tmp && EDMP_PMC_FORWARDS=12 rocprofv3 --kernel-trace --output-format csv -d $d -o p -- python $REPO/scripts/pmc_unet_forward.py > /dev/null 2>&1)
  python - "$lib" <<PY
import csv,glob,sys
f=glob.glob("/tmp/ab_out/**/p_kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
rows=[r for r in rows if 'time_table' not in r['Kernel_Name']]
per=len(rows)//12
rows=rows[2*per:]            # drop two warm-up forwards
import collections
agg=collections.OrderedDict(); tot=0
for r in rows:
    n=r['Kernel_Name'].split('(')[0].replace('edmp::','').replace('void ','').replace('_kernel','')
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000
    a=agg.setdefault(n,[0,0.0]); a[0]+=1; a[1]+=d; tot+=d
span=(int(rows[-1]['End_Timestamp'])-int(rows[0]['Start_Timestamp']))/1000/10
fam=collections.OrderedDict()
for n,(c,t) in agg.items():
    k=n.split('<')[0]; fam[k]=fam.get(k,0)+t/10
print(f"{sys.argv[1]:28s} kernels/fwd {tot/10:7.1f} us  span/fwd {span:7.1f} | "+"  ".join(f"{k}={v:.0f}" for k,v in fam.items())+" | "+"  ".join(f"{n.replace('rcb_','')}={t/c:.1f}" for n,(c,t) in agg.items() if n.startswith(('rcb_conv','rcb_rows<64'))))
PY
done
done
cp /tmp/orig.so edmp_amd/libedmp_hip.so
