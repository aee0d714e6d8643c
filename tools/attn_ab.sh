#!/bin/bash
# Same-box A/B of the prefill attention kernel variants (tools/attn_only.py); usage: tools/attn_ab.sh <outfile>
out=${1:-gpurun_out/attn_ab.log}
: > $out
run() { echo "--- $*" >> $out; env "$@" python tools/attn_only.py $SHAPE >> $out 2>&1; }
for SHAPE in whisper beats beats_bias llama; do
  export SHAPE
  [ -f icl-speech-text-llm_amd/lib/libicl_hip_base.so ] && run ICL_LIB=icl-speech-text-llm_amd/lib/libicl_hip_base.so
  run ICL_ATTN_PIPE=0 ICL_ATTN_NW=4
  run ICL_ATTN_PIPE=0 ICL_ATTN_NW=8
  run ICL_ATTN_PIPE=1 ICL_ATTN_NW=4
  run ICL_ATTN_PIPE=1 ICL_ATTN_NW=8
done
cat $out
