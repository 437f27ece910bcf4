// Absolute FP64 throughput of gfx950 by wall clock (hipEvents): MFMA f64 16x16x4 and VALU FMA at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define ITERS 4096
template <int MODE>  // 0: MFMA 4 acc; 1: FMA 8 chains; 2: MFMA + FMA interleaved
__global__ void k(double* out, double seed) {
    d4 c[4]; for (int i = 0; i < 4; ++i) c[i] = d4{seed, seed, seed, seed};
    double a[8]; for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x * 1e-6;
    double x = seed * 0.999999, y = seed * 1e-9;
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c[i], 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = fma(a[i], x, y);
        }
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int threads, int wgs) {
    double* out; (void)hipMalloc(&out, size_t(wgs) * threads * sizeof(double));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(threads), 0, 0, out, 1.0000001);
    hipEventRecord(e0); const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(threads), 0, 0, out, 1.0000001);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    double waves = double(wgs) * threads / 64.0;
    double flop = 0;
    if (MODE == 0 || MODE == 2) flop += waves * ITERS * 4 * 2048.0;
    if (MODE == 1 || MODE == 2) flop += waves * ITERS * 32 * 64 * 2.0;
    printf("%-34s WG %4d x %5d WGs: %8.3f ms  %7.2f TFLOP/s\n", name, threads, wgs, ms, flop / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}
int main() {
    for (int t : {256, 512, 1024}) {
        run<0>("MFMA f64 16x16x4", t, 256);
        run<0>("MFMA f64 16x16x4", t, 1024);
        run<1>("VALU FP64 FMA", t, 256);
        run<1>("VALU FP64 FMA", t, 1024);
        run<2>("MFMA + VALU FMA same wave", t, 256);
    }
    return 0;
}
