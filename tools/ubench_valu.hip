// How fast does ONE wave issue instructions on this chip?  (sampler design input)  dependent / independent VALU chains,
// v_cmp -> s_bcnt1 (VALU -> SALU), ds_bpermute chains; 1 block, 1 wave (and 16 waves for comparison).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int N = 4096;
__global__ void k_dep(float* o, float a) { float x = o[threadIdx.x]; long long t0 = clock64();
#pragma unroll 64
  for (int i = 0; i < N; ++i) x = __builtin_fmaf(x, a, 1.0f);
  long long t1 = clock64(); o[threadIdx.x] = x; if (threadIdx.x == 0) ((long long*)o)[64] = t1 - t0; }
__global__ void k_ind(float* o, float a) { float x0 = o[threadIdx.x], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; long long t0 = clock64();
#pragma unroll 8
  for (int i = 0; i < N / 8; ++i) { x0 = __builtin_fmaf(x0, a, 1.0f); x1 = __builtin_fmaf(x1, a, 1.0f); x2 = __builtin_fmaf(x2, a, 1.0f); x3 = __builtin_fmaf(x3, a, 1.0f);
    x4 = __builtin_fmaf(x4, a, 1.0f); x5 = __builtin_fmaf(x5, a, 1.0f); x6 = __builtin_fmaf(x6, a, 1.0f); x7 = __builtin_fmaf(x7, a, 1.0f); }
  long long t1 = clock64(); o[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; if (threadIdx.x == 0) ((long long*)o)[64] = t1 - t0; }
__global__ void k_cmp(float* o, unsigned a) { unsigned x = ((unsigned*)o)[threadIdx.x]; int c = 0; long long t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) c += __popcll(__builtin_amdgcn_uicmp(x, a + i, 34));
  long long t1 = clock64(); o[threadIdx.x] = (float)c; if (threadIdx.x == 0) ((long long*)o)[64] = t1 - t0; }
__global__ void k_perm(float* o) { int x = threadIdx.x; long long t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N / 4; ++i) x = __shfl_xor(x, 17, 64) + 1;
  long long t1 = clock64(); o[threadIdx.x] = (float)x; if (threadIdx.x == 0) ((long long*)o)[64] = t1 - t0; }
template <typename F> static int run(const char* name, int threads, int n_ops, F launch, float* d) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int r = 0; r < 20; ++r) launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); long long cyc; CK(hipMemcpy(&cyc, (long long*)d + 64, 8, hipMemcpyDeviceToHost));
  printf("%-28s %4d threads: %7.2f us/kernel, %8lld clock64 ticks -> %.2f ticks/op, %.2f ns/op (kernel incl. launch)\n", name, threads, ms * 1e3 / 20, cyc, (double)cyc / n_ops, ms * 1e6 / 20 / n_ops);
  return 0; }
int main() { float* d; CK(hipMalloc(&d, 1 << 16)); CK(hipMemset(d, 0, 1 << 16));
  for (int th : {64, 1024}) {
    run("dependent v_fma chain", th, N, [&] { hipLaunchKernelGGL(k_dep, dim3(1), dim3(th), 0, 0, d, 1.0001f); }, d);
    run("8 independent v_fma chains", th, N, [&] { hipLaunchKernelGGL(k_ind, dim3(1), dim3(th), 0, 0, d, 1.0001f); }, d);
    run("v_cmp -> s_bcnt1 -> s_add", th, N, [&] { hipLaunchKernelGGL(k_cmp, dim3(1), dim3(th), 0, 0, d, 12345u); }, d);
    run("dependent ds_bpermute", th, N / 4, [&] { hipLaunchKernelGGL(k_perm, dim3(1), dim3(th), 0, 0, d); }, d);
  }
  return 0; }
