#!/bin/bash
# Multi-GPU scaling run on ONE node: bench.py at N = 1, 2, 4, 8 (or the GPUs present), one rank per GPU over RCCL, exactly as
# the driver launches it.  Prints one JSON line per N.  Usage: scripts/run_scale.sh [extra bench.py flags, e.g. --logical-batch
# --guides 1,2,3,4,5,10,11,13]
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
STEPS=${EDMP_SCALE_STEPS:-3}
WARM=${EDMP_SCALE_WARMUP:-1}
PORT=${EDMP_SCALE_PORT:-29611}
for N in 1 2 4 8; do
  if [ "$N" -gt "$NGPU" ]; then echo "{\"n_gpus\": $N, \"skipped\": \"only $NGPU GPU(s) visible\"}"; continue; fi
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-cpu-baseline --no-roofline "$@" | tail -1
  else
    # torch.distributed.run exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; bench.py binds rank r to cuda:LOCAL_RANK
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((PORT + N)) \
      bench.py --gpus $N --steps $STEPS --warmup $WARM --no-cpu-baseline --no-roofline "$@" | tail -1
  fi
done
