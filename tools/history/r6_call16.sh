#!/bin/sh
# round 6, call 16: write-through (sc1) output stores in the mixer's three kernels, one at a time and together
sh tools/tm_store_ab.sh tms2 gup gres gboth gall
