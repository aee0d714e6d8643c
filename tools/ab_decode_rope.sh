#!/bin/bash
for round in 1 2 3; do
  for f in 1 0; do
    echo -n "round $round fuse=$f batch1: "; ICL_FUSE_DECODE_ROPE=$f python bench.py --batch 1 --steps 40 --warmup 2 --no-cpu-baseline --no-through-plugin --no-other-workloads --no-phases 2>&1 | grep "timed region"
  done
done
for round in 1 2; do
  for f in 1 0; do
    echo -n "round $round fuse=$f batch256: "; ICL_FUSE_DECODE_ROPE=$f python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-through-plugin --no-other-workloads 2>&1 | grep -E "phases:"
  done
done
