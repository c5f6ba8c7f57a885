#!/bin/sh
# round 6, call 12: FETCH / WRITE counters of the config-3 forward (short form), the new objects of bench.py's config-4 leg, new route tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_config45_gpu.py tests/test_config3_gpu.py -q 2>&1 | tail -3
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pr_*
C3P="python $R/bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-stage-profile --no-extras"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pr_f_c3 -o p -- $C3P > $O/r6c12_pmc_f.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pr_w_c3 -o p -- $C3P > $O/r6c12_pmc_w.log 2>&1; echo "write rc=$?"
python $R/tools/pmc_to_json.py $O/r6c12_pmc_traffic_config3.json /tmp/pr_f_c3 /tmp/pr_w_c3
cd $R; timeout 900 python bench.py --leg config4 > $O/r6c12_config4.json 2> $O/r6c12_config4.err; echo "c4 rc=$?"; tail -c 2500 $O/r6c12_config4.json
