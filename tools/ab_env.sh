#!/bin/bash
# in-situ A/B of an environment switch: tools/ab_env.sh <outfile> VAR a b [rounds]  -> the bench's own step per value, interleaved
out=$1; var=$2; a=$3; b=$4; rounds=${5:-2}
: > $out
for round in $(seq $rounds); do
  for v in $a $b; do
    env $var=$v python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-through-plugin --no-other-workloads 2> /tmp/ab_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$round $var=$v', d['value'], d['ms_per_step'], 'enc', d['phases']['encoder']['ms'], 'pre', d['phases']['prefill']['ms'], 'dec', d['phases']['decode']['ms'], 'tokens', d['first_utterance_tokens'][:3])" >> $out
  done
done
cat $out
