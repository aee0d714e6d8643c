#!/bin/bash
# Same-box A/B of prefill-attention builds / variants (tools/attn_only.py).  usage: tools/attn_ab.sh <outfile> [shapes...]
out=${1:-gpurun_out/attn_ab.log}; shift
shapes=${@:-whisper beats}
: > $out
for SHAPE in $shapes; do
  for lib in icl-speech-text-llm_amd/lib/libicl_hip*.so; do
    for il in 0 1; do
      echo "--- $SHAPE $lib ICL_ATTN_IL=$il" >> $out
      ICL_ATTN_IL=$il ICL_LIB=$lib python tools/attn_only.py $SHAPE 2>&1 | grep "attn " >> $out
    done
  done
done
cat $out
