#!/bin/bash
# After `ROUND=rNN scripts/gpu_round.sh` ran on a GPU box (gpurun merges gpurun_out/rNN/ back): copy that run's summaries into profiles/
# (tracked) WITHOUT touching the raw output, and derive the numbers profiles/README.md quotes from the committed files.
R=${1:-r05}
O=gpurun_out/$R
set -e
cp $O/bench.json profiles/${R}_bench.json
for c in c2 c5_n1 c5_n1_pyhook; do [ -s $O/bench_$c.json ] && grep '^{' $O/bench_$c.json | tail -1 > profiles/${R}_bench_$c.json; done  # (RCCL prints its version banner on stdout: the JSON line is the one that starts with a brace)
python - <<EOF
import json
b = json.load(open('$O/bench.json'))
if 'problem_set' in b:
    json.dump(b['problem_set'], open('profiles/${R}_problem_set.json', 'w'), indent=1)
EOF
cp $O/kernel_stats.csv profiles/${R}_kernel_stats.csv
cp $O/profile_summary.json profiles/${R}_profile_summary.json
cp $O/pmc_hbm_traffic.json profiles/${R}_pmc_hbm_traffic.json
cp $O/pmc_mfma_util.md profiles/${R}_pmc_mfma_util.md
python scripts/round_numbers.py $R > profiles/${R}_numbers.md
cat profiles/${R}_numbers.md
