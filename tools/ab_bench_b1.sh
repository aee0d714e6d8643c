#!/bin/bash
# in-situ A/B of library builds at ONE utterance per step (the reference CLI's --batch_size 1): seconds for 40 utterances, three interleaved rounds
for round in 1 2 3; do
  for lib in "$@"; do
    echo -n "$round $lib: "; ICL_LIB_PATH=$PWD/icl-speech-text-llm_amd/lib/$lib python bench.py --batch 1 --steps 40 --warmup 2 --no-cpu-baseline --no-through-plugin --no-other-workloads --no-phases 2>&1 | grep "timed region" | sed 's/.*timed region: //'
  done
done
