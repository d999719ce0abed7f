#!/bin/bash
# same-box A/B of the bf16x3 layers (EDMP_BF16X3 mask) + the GPU parity subset that covers them
O=gpurun_out/${ROUND:-r06e}
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -s -k "${K:-unet_golden or teacher_forced_steps or karatsuba or bf16x3 or ragged or fused_and_unfused or sixteen or packed}" > $O/pytest.log 2>&1; grep -E "^\[bf16x3|passed|failed|Error" $O/pytest.log | cut -c1-260 | tail -${TAILN:-30}
for v in ${MASKS:-0x7 0}; do EDMP_BF16X3=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-two-scenes --no-problem-set --no-native-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BF16X3=$v', round(d['value']), round(d['ms_per_step'],2))"; done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-two-scenes --no-problem-set > $O/bench.json 2> $O/bench.err; python scripts/show_bench.py $O/bench.json | cut -c1-250
