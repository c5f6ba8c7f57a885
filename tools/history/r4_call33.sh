#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python bench.py > $O/c33_bench.json 2> $O/c33_bench.err
echo "bench rc=$?"
tail -c 1500 $O/c33_bench.json
timeout 1500 sh tools/profile_round.sh r4b > $O/c33_profile.log 2>&1
echo "profile rc=$?"
ls $O | grep prof_r4b
