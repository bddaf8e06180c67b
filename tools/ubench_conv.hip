// Phase profile of the vocoder's wide plane-input conv kernel (k_conv1d_bf3p): cycles of wave 0 per phase, summed over blocks.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFS_C3_PROF -I fish-speech.rs_amd/csrc tools/ubench_conv.hip -o tools/ubench_conv.bin
// run:   tools/ubench_conv.bin [C=128] [T=65536] [K=7] [dil=3] [f16=1]
#include "codec_conv_bf3.hip"

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 128, T = argc > 2 ? atoi(argv[2]) : 65536, K = argc > 3 ? atoi(argv[3]) : 7, dil = argc > 4 ? atoi(argv[4]) : 3;
    const bool f16 = argc > 5 ? atoi(argv[5]) != 0 : true;
    const size_t plane_elems = (size_t)2 * (C / 8) * (fs::CODEC_PLANE_PAD + T) * 8 + (64 << 10);
    uint16_t *xp, *yp, *wp; float *bias, *res, *y;
    CK(hipMalloc(&xp, plane_elems * 2)); CK(hipMalloc(&yp, plane_elems * 2));
    CK(hipMemset(xp, 0, plane_elems * 2));
    const size_t wn = fs::codec_pack_bf3_elems(C, K, C, f16);
    CK(hipMalloc(&wp, wn * 2)); CK(hipMemset(wp, 0, wn * 2));
    CK(hipMalloc(&bias, C * 4)); CK(hipMemset(bias, 0, C * 4));
    CK(hipMalloc(&res, (size_t)C * T * 4)); CK(hipMemset(res, 0, (size_t)C * T * 4));
    CK(hipMalloc(&y, (size_t)C * T * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {  // 0: conv1 (planes out only), 1: conv2 (residual, f32 + planes out)
        auto run = [&]() {
            fs::codec_conv1d_bf3(nullptr, xp, 1, C, T, wp, f16, bias, C, K, dil, true, mode ? fs::CODEC_EPI_RES : fs::CODEC_EPI_NONE, mode ? res : nullptr, nullptr,
                                 mode ? y : nullptr, yp, true, 1, st);
        };
        for (int i = 0; i < 3; ++i) run();
        CK(hipStreamSynchronize(st));
        unsigned long long z[8] = {};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(fs::g_c3prof), z, sizeof(z)));
        const int reps = 20;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) run();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long pr[8];
        CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(fs::g_c3prof), sizeof(pr)));
        double tot = 0; for (int i = 0; i < 6; ++i) tot += (double)pr[i];
        printf("%s C=%d T=%d K=%d dil=%d %s: %.1f us per conv; wave-0 cycles by phase: wait-prev-barrier %.1f%%  issue-DMA %.1f%%  vmcnt(0) %.1f%%  barrier %.1f%%  "
               "ds_read+MFMA %.1f%%  epilogue %.1f%%   (%.0f cycles per block)\n", f16 ? "f16" : "bf16x3", C, T, K, dil, mode ? "conv2 (res, f32+planes out)" : "conv1 (planes out)",
               ms * 1e3 / reps, 100 * pr[0] / tot, 100 * pr[1] / tot, 100 * pr[2] / tot, 100 * pr[3] / tot, 100 * pr[4] / tot, 100 * pr[5] / tot,
               tot / reps / ((double)((T + 255) / 256) * (C / 32)));
    }
    return 0;
}
