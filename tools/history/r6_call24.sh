#!/bin/sh
# round 6, call 24: what the driver runs at round end, on the final code: smoke(), the default bench line, the GPU suite
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py --no-extras 2>/dev/null | tail -c 600
