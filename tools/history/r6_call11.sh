#!/bin/sh
# round 6, call 11: GPU suite on track.hip without IEEE-mode canonicalisation; config 3 with the tiled bf16 gather forced on / off; the
# FETCH / WRITE counter passes of the config-3 leg (with their logs this time)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/r6c11_pytest_gpu.log 2>&1
echo "tests rc=$?"; tail -4 $O/r6c11_pytest_gpu.log
sh tools/ab_c3.sh PIPS_GATHER_TILED 0 1
export TMPDIR=/tmp; cd /tmp
C3="python $R/bench.py --leg config3"
rm -rf /tmp/pr_*
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pr_f_c3 -o p -- $C3 > $O/r6c11_pmc_f.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pr_w_c3 -o p -- $C3 > $O/r6c11_pmc_w.log 2>&1; echo "write rc=$?"
find /tmp/pr_f_c3 -type f | head -5
python $R/tools/pmc_to_json.py $O/r6c11_pmc_traffic_config3.json /tmp/pr_f_c3 /tmp/pr_w_c3
grep -v "^W2026\|^I2026" $O/r6c11_pmc_f.log | tail -5 | cut -c1-300
