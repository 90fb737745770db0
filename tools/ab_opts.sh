#!/bin/bash
# usage: tools/ab_opts.sh <option> "<v1 v2 ...>" <rounds> [bench args] -- alternates the values of one library option, <rounds> times (dev helper)
opt=$1; vals=$2; rounds=$3; shift 3
for r in $(seq $rounds); do for v in $vals; do
  echo -n "$opt=$v: "; timeout 300 python tools/bench_brief.py --steps 300 --warmup 20 --opt $opt=$v "$@" | sed -e 's/.*| views/views/' | cut -c1-40,290-420
done; done
