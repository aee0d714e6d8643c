#!/bin/bash
# in-situ A/B of library builds: the bench's own step (3 timed steps + phase step), two interleaved rounds
out=$1; shift
: > $out
for round in 1 2; do
  for lib in "$@"; do
    ICL_LIB_PATH=$PWD/icl-speech-text-llm_amd/lib/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-through-plugin --no-other-workloads 2> /tmp/ab_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$round $lib', d['value'], d['ms_per_step'], 'enc', d['phases']['encoder']['ms'], 'pre', d['phases']['prefill']['ms'], 'dec', d['phases']['decode']['ms'], 'frac', d['roofline']['frac'])" >> $out
  done
done
cat $out
