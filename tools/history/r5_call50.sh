#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python bench.py > $O/r5_bench.json 2> $O/r5_bench.err
echo "bench rc=$?"; tail -c 200 $O/r5_bench.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
