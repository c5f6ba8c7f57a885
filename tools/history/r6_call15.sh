#!/bin/sh
# round 6, call 15: do the dirty lines a kernel leaves in the L2s cost it time at its end?  token-mix stores with write-through / nontemporal policies
sh tools/tm_store_ab.sh
