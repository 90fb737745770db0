#!/bin/bash
# usage: tools/ab_binning.sh [bench args] -- same-box A/B of the two binning schemes
for b in 0 1 0 1; do timeout 200 python tools/bench_brief.py --steps 30 --warmup 5 --binning $b "$@"; done
