#!/bin/bash
# A/B of library variants under tools/_ab/lib*.so (scratch, git-ignored) on the GPU box: greedy us/frame at configs[1] shapes.
for v in "$@"; do
  cp tools/_ab/lib$v.so fish-speech.rs_amd/libfishrt.so
  for d in bf16 fp8; do echo -n "$v "; timeout 150 python tools/p2_quick.py $d 2>&1 | tail -1; done
done
