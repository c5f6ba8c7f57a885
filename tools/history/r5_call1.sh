#!/bin/sh
# round 5, call 1: full GPU suite on the cheap closures + the bench line with the launch-train roofline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/r5c1_tests.log 2>&1
echo "tests rc=$?"; tail -5 $O/r5c1_tests.log
timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/r5c1_bench.json 2> $O/r5c1_bench.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5c1_bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"]); r=d["roofline"]
print({k:r[k] for k in ("frac","launch_ms","kernel_body_ms","launch_ms_rocprof","frac_rocprof","kernel")})
print({k:(v["ms"],v["kernel_body_ms"]) for k,v in r["all"].items()})
print(d["stages_ms"])
PY
