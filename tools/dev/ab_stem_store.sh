#!/bin/bash
# dev A/B (same gpurun call, alternating): the fused VGG stem with 8-byte (this build) vs conflict-free 16-byte (tools/dev/ab/lib_stem_wide.so =
# build_variant.sh vgg_stem2.hip stem_wide -DS2_WIDE_STORE) LDS stores of its conv1_1 epilogue; embedder launch by launch, lists off + on
for i in 1 2 3; do
  echo "wide:   $(CVPCE_LIB=tools/dev/ab/lib_stem_wide.so timeout -k 10 200 python tools/dev/embed_layers.py 2>&1 | grep -E 'skip=|stem' | tr '\n' '|' | cut -c1-400)"
  echo "narrow: $(timeout -k 10 200 python tools/dev/embed_layers.py 2>&1 | grep -E 'skip=|stem' | tr '\n' '|' | cut -c1-400)"
done
