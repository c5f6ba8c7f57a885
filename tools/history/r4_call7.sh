#!/bin/sh
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pkn -o p -- python $R/tools/blas_kernel_names.py > /dev/null 2>&1
for f in $(find /tmp/pkn -name "*.db"); do python - "$f" <<'PY' > $R/gpurun_out/r4_call7_blas_kernels.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for name, calls, total, avg, pct in db.execute("select * from top_kernels"):
    print("%8d calls %10.2f us avg  %s" % (calls, avg, name))
PY
done
cat $R/gpurun_out/r4_call7_blas_kernels.txt | cut -c1-400
