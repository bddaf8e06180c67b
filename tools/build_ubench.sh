#!/bin/bash
# builds libfishrt.so and the kernel micro-benchmarks (tools/*.bin, git-ignored)
set -e
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
bash "$R/fish-speech.rs_amd/build.sh" | tail -1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I "$R/fish-speech.rs_amd/csrc" -c "$R/tools/ubench_lm.hip" -o /tmp/ubench_lm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/ubench_lm.o "$R/fish-speech.rs_amd/build/lm_kernels.o" -o "$R/tools/ubench_lm.bin"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I "$R/fish-speech.rs_amd/csrc" -c "$R/tools/ubench_gemm.hip" -o /tmp/ubench_gemm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/ubench_gemm.o "$R/fish-speech.rs_amd/build/lm_kernels.o" -o "$R/tools/ubench_gemm.bin"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I "$R/fish-speech.rs_amd/csrc" -c "$R/tools/ubench_sample.hip" -o /tmp/ubench_sample.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/ubench_sample.o "$R/fish-speech.rs_amd/build/lm_kernels.o" -o "$R/tools/ubench_sample.bin"
