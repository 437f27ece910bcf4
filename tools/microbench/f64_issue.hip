// Micro-benchmarks that ground the kernel design: FP64 issue rate / latency on gfx950 for the instruction
// mixes the MPC kernel uses.  hipcc --offload-arch=gfx950 -O3 f64_issue.hip -o f64_issue && ./f64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
#define N_IT 256

__device__ __forceinline__ double readlane_f64(double x, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// mode 0: 8 independent FMA chains; 1: one dependent chain; 2: readlane+fma (independent accumulators);
// 3: rsq chain; 4: MFMA 4 independent accumulators; 5: MFMA dependent; 6: MFMA(4 acc) + 8 independent FMAs per MFMA;
// 7: FMA only, same count as in 6 (reference); 8: ds_read_b64 broadcast + fma
template <int MODE>
__global__ void bench(double* out, unsigned long long* cyc, double seed) {
    __shared__ double lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = seed + i;
    __syncthreads();
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3 + i;
    d4 c[4];
    for (int i = 0; i < 4; ++i) c[i] = d4{seed, seed, seed, seed};
    double x = seed * 0.5, y = seed * 0.25;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N_IT; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = fma(a[i], x, y);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[0] = fma(a[0], x, y);
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = fma(readlane_f64(a[(i + 1) & 7], i + 3), x, a[i]);
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[0] = __builtin_amdgcn_rsq(a[0]) + y;
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c[i & 3], 0, 0, 0);
        } else if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c[0], 0, 0, 0);
        } else if (MODE == 6) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                c[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c[i & 3], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = fma(a[j], x, y);
            }
        } else if (MODE == 7) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = fma(a[j], x, y);
            }
        } else if (MODE == 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = fma(lds[(it * 8 + i) & 1023], x, a[i]);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, double per_iter_instr) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(double)); hipMalloc(&cyc, 256 * sizeof(unsigned long long));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(bench<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0000001);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-58s threads/WG %4d : %8.1f cycles per instruction-unit (median WG)\n", name, threads,
           double(h[128]) / (N_IT * per_iter_instr));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int threads : {256, 512}) {
        run<0>("FP64 FMA, 8 independent chains [per FMA]", threads, 8);
        run<1>("FP64 FMA, dependent chain [per FMA]", threads, 8);
        run<2>("readlane(2x b32) + FP64 FMA [per pair]", threads, 8);
        run<3>("v_rsq_f64 + add dependent [per rsq+add]", threads, 8);
        run<4>("MFMA f64 16x16x4, 4 independent acc [per MFMA]", threads, 8);
        run<5>("MFMA f64 16x16x4, dependent acc [per MFMA]", threads, 8);
        run<6>("MFMA + 8 independent FP64 FMA [per MFMA+8FMA group]", threads, 8);
        run<7>("8 independent FP64 FMA only [per 8-FMA group]", threads, 8);
        run<8>("ds_read_b64 (broadcast) + FMA [per pair]", threads, 8);
    }
    return 0;
}
