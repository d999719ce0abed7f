#!/bin/bash
# 8-GPU-day checklist on ONE node: everything the first multi-GPU lease has to produce, in one run, one JSON out.
#   1. the RCCL two-rank test (tests/test_gpu_parity.py::test_rccl_two_ranks: sharded loop over a real all-reduce, native hook included)
#   2. C4: bench.py replicas at N = 1, 2, 4, 8, launched exactly as the driver launches it (one rank per GPU over RCCL)
#   3. C5: bench.py --logical-batch with the eight-guide ensemble at the same N: per-guided-step all-reduce from inside the device loop
#      (native ncclAllReduce hook over RCCL; the Python callback over gloo), hook host time, the gathered best row
# Usage: scripts/run_scale.sh [out.json]            real run: N up to the GPUs visible, RCCL
#        DRY=1 scripts/run_scale.sh [out.json]      dry run on a ONE-GPU box: the same launch lines with EDMP_DIST_BACKEND=gloo (ranks share
#                                                   the GPU), small batches - exercises ports, rendezvous, re-exec, JSON plumbing, not speed
# Knobs: EDMP_SCALE_STEPS (3), EDMP_SCALE_WARMUP (1), EDMP_SCALE_PORT (29611), EDMP_SCALE_NS ("1 2 4 8").
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=${1:-gpurun_out/scale.json}
mkdir -p "$(dirname "$OUT")"
DRY=${DRY:-0}
NGPU=$(python -c 'import torch; print(torch.cuda.device_count())')
STEPS=${EDMP_SCALE_STEPS:-3}
WARM=${EDMP_SCALE_WARMUP:-1}
PORT=${EDMP_SCALE_PORT:-29611}
NS=${EDMP_SCALE_NS:-"1 2 4 8"}
EXTRA="--no-cpu-baseline --no-roofline --no-two-scenes --no-problem-set --no-native-leg"
if [ "$DRY" = 1 ]; then
  export EDMP_DIST_BACKEND=gloo
  EXTRA="$EXTRA --batch 64"
  STEPS=1; WARM=0
fi
TMP=$(mktemp -d)
echo "[scale] $NGPU GPU(s) visible, dry=$DRY, N in {$NS}" >&2

# 1. RCCL between two devices (skips itself with the reason on a one-GPU box)
python -m pytest tests/test_gpu_parity.py -q -x -k "test_rccl_two_ranks or test_rccl_branch_world_size_one" -rs > $TMP/rccl.txt 2>&1
echo "rc=$?" >> $TMP/rccl.txt

run_bench() {  # $1 = N, $2 = tag, rest = bench flags
  local N=$1 TAG=$2; shift 2
  if [ "$DRY" != 1 ] && [ "$N" -gt "$NGPU" ]; then echo "{\"n_gpus\": $N, \"skipped\": \"only $NGPU GPU(s) visible\"}" > $TMP/${TAG}_$N.json; return; fi
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 --steps $STEPS --warmup $WARM $EXTRA "$@" 2> $TMP/${TAG}_$N.err | grep '^{' | tail -1 > $TMP/${TAG}_$N.json
  else
    # the driver's launch line; torch.distributed.run exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; bench.py binds rank r to cuda:LOCAL_RANK
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((PORT + N)) \
      bench.py --gpus $N --steps $STEPS --warmup $WARM $EXTRA "$@" 2> $TMP/${TAG}_$N.err | grep '^{' | tail -1 > $TMP/${TAG}_$N.json
  fi
  [ -s $TMP/${TAG}_$N.json ] || echo "{\"n_gpus\": $N, \"error\": \"no JSON line; stderr tail: $(tail -3 $TMP/${TAG}_$N.err | tr '\n"' ' .' | cut -c1-300)\"}" > $TMP/${TAG}_$N.json
}
for N in $NS; do
  run_bench $N c4
  run_bench $N c5 --guides 1,2,3,4,5,10,11,13 --logical-batch
done

python - "$TMP" "$OUT" "$DRY" "$NGPU" $NS <<'PY'
import json, sys
tmp, out, dry, ngpu, ns = sys.argv[1], sys.argv[2], sys.argv[3] == "1", int(sys.argv[4]), [int(n) for n in sys.argv[5:]]
def line(tag, n):
    try:
        d = json.load(open(f"{tmp}/{tag}_{n}.json"))
    except Exception as e:  # noqa: BLE001
        return {"n_gpus": n, "error": repr(e)}
    if "value" not in d:
        return d
    keep = {k: d.get(k) for k in ("n_gpus", "n_ranks_seen", "dist_backend", "value", "unit", "ms_per_step", "steps", "scaling", "allreduce_hook")}
    keep["global_batch"] = d["config"]["global_batch"]
    keep["parallelism"] = d["config"]["parallelism"]
    keep["best"] = d.get("best")
    return keep
rccl = open(f"{tmp}/rccl.txt").read()
res = {"dry_run_over_gloo_on_one_gpu": dry, "gpus_visible": ngpu,
       "rccl_tests": {"tail": rccl.strip().splitlines()[-6:], "two_ranks": ("skipped" if "SKIPPED" in rccl or "skipped" in rccl.split("rc=")[0].splitlines()[-1] else "ran")},
       "c4_replicas": [line("c4", n) for n in ns], "c5_logical_batch_eight_guides": [line("c5", n) for n in ns]}
base = next((r["value"] for r in res["c4_replicas"] if r.get("n_gpus") == 1 and "value" in r), None)
if base:
    for r in res["c4_replicas"]:
        if "value" in r:
            r["vs_n1_per_gpu"] = r["value"] / r["n_gpus"] / base  # (weak scaling: per-GPU rate against N = 1; informative - the driver computes its own)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: (v if k != "rccl_tests" else v["two_ranks"]) for k, v in res.items()})[:1500])
PY
rm -rf $TMP
