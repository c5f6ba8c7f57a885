#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 sh tools/profile_mfma.sh r4 > $O/c34_mfma.log 2>&1
echo "mfma rc=$?"
head -20 $O/r4_pmc_mfma_util_exact.txt
timeout 900 python bench.py > $O/c34_bench.json 2> $O/c34_bench.err
echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/c34_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["kernel"])
PY
