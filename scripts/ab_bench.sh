#!/bin/bash
# bench.py (no roofline / CPU legs) with every library under ${EDMP_AB_DIR:-scratch/ab}/*.so in turn, twice, on ONE box
cp edmp_amd/libedmp_hip.so /tmp/orig.so
for rep in 1 2; do
for lib in ${EDMP_AB_DIR:-scratch/ab}/*.so; do
  cp $lib edmp_amd/libedmp_hip.so
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']), round(d['ms_per_step'],2), d['end_to_end_scene_seconds']['numpy_stream_drawn_and_uploaded_per_scene'])"
done
done
cp /tmp/orig.so edmp_amd/libedmp_hip.so
