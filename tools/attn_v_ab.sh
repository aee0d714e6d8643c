#!/bin/bash
# Same-box A/B of attention builds: every lib/libicl_hip*.so, shapes llama (128 seqs) / beats_bias / whisper, 3 rounds interleaved,
# plus a bit comparison of each variant against the default build.  usage: tools/attn_v_ab.sh <outdir>
out=${1:-gpurun_out/attn_v_ab}; mkdir -p $out; : > $out/times.log
base=icl-speech-text-llm_amd/lib/libicl_hip.so
ICL_LIB=$base python tools/attn_bitcmp.py $out/base.pt || exit 1
for lib in icl-speech-text-llm_amd/lib/libicl_hip_*.so; do
  n=$(basename $lib .so)
  ICL_LIB=$lib python tools/attn_bitcmp.py $out/$n.pt && python tools/attn_bitcmp.py --cmp $out/base.pt $out/$n.pt >> $out/times.log 2>&1
done
for round in 1 2 3; do
  for lib in icl-speech-text-llm_amd/lib/libicl_hip*.so; do
    for shape in llama beats_bias whisper; do
      nseq=64; [ $shape = llama ] && nseq=128
      echo -n "$round $(basename $lib) " >> $out/times.log
      ICL_ATTN_NSEQ=$nseq ICL_LIB=$lib python tools/attn_only.py $shape 2>&1 | grep "attn " >> $out/times.log
    done
  done
done
rm -f $out/*.pt
cat $out/times.log
