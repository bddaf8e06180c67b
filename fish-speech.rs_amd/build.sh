#!/bin/bash
# Builds libfishrt.so (HIP kernels + C ABI) for gfx950, in-tree.  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
cd "$HERE/csrc"
mkdir -p "$HERE/build"
# -amdgpu-kernarg-preload-count: the CP hands the first kernel arguments to the wave in SGPRs at launch (saves the s_load round trip at
# the top of every graph node: -1.3% on the 266-node decode frame, measured with tools/ubench_lm)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=on -mllvm -amdgpu-kernarg-preload-count=16"
pids=()
for f in lm_kernels.hip lm_persist.hip lm_persist_slow.hip lm_persist_rows.hip lm_engine.hip codec_kernels.hip codec_conv_bf3.hip codec_engine.hip fs_comm.cpp fishrt_api.cpp; do
  o="$HERE/build/${f%.*}.o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ -n "$(find . -name '*.h' -newer "$o" -print -quit)" ] || [ "../../include/fishrt.h" -nt "$o" ]; then
    # (lm_persist_rows.hip: four unrolled layers x two row groups x R rows exceed clang's default budget for `#pragma unroll`; a loop it refuses to
    # unroll indexes the resident weight-fragment array at run time, which moves the whole array to scratch)
    EXTRA=""; [ "$f" = lm_persist_rows.hip ] && EXTRA="-mllvm -pragma-unroll-threshold=1000000"
    ( /opt/rocm/bin/hipcc $FLAGS $EXTRA -x hip -c "$f" -o "$o" ) &
    pids+=($!)
  fi
done
# the range-counting twin of the vocoder's f16 conversion kernels (csrc/codec_conv_bf3.hip compiled again with -DFS_C3_CHECK into fs::c3chk:
# fs_codec_set_range_check); same source, second object
o="$HERE/build/codec_conv_bf3_chk.o"
if [ ! -f "$o" ] || [ codec_conv_bf3.hip -nt "$o" ] || [ -n "$(find . -name '*.h' -newer "$o" -print -quit)" ]; then
  ( /opt/rocm/bin/hipcc $FLAGS -DFS_C3_CHECK -x hip -c codec_conv_bf3.hip -o "$o" ) &
  pids+=($!)
fi
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$HERE"/build/*.o -o "$HERE/libfishrt.so"
echo "built $HERE/libfishrt.so"
