// Accuracy of v_rsq_f64 / v_rcp_f64 seeds and of 1 / 2 Newton steps on gfx950 (decides the step count in fast_rsqrt).
// build: hipcc --offload-arch=gfx950 -O3 -o rsq_accuracy rsq_accuracy.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

__global__ void k(const double* d, double* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = d[i];
    double y0 = __builtin_amdgcn_rsq(x);
    double h = 0.5 * x;
    double y1 = y0 * fma(-h * y0, y0, 1.5);
    double y2 = y1 * fma(-h * y1, y1, 1.5);
    double r0 = __builtin_amdgcn_rcp(x);
    double r1 = r0 * fma(-x, r0, 2.0);
    double r2 = r1 * fma(-x, r1, 2.0);
    o[6 * i] = y0; o[6 * i + 1] = y1; o[6 * i + 2] = y2; o[6 * i + 3] = r0; o[6 * i + 4] = r1; o[6 * i + 5] = r2;
}

int main() {
    const int n = 1 << 20;
    std::vector<double> h(n), o(6 * n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-10.0, 20.0);
    for (auto& v : h) v = std::exp2(u(g));
    double *d, *dout;
    hipMalloc(&d, n * 8); hipMalloc(&dout, 6 * n * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, dout, n);
    hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
    long double e[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        long double rs = 1.0L / sqrtl((long double)h[i]), rc = 1.0L / (long double)h[i];
        for (int j = 0; j < 3; ++j) e[j] = fmaxl(e[j], fabsl((o[6 * i + j] - rs) / rs));
        for (int j = 3; j < 6; ++j) e[j] = fmaxl(e[j], fabsl((o[6 * i + j] - rc) / rc));
    }
    printf("max relative error over %d samples (eps = 2.2e-16)\n", n);
    printf("rsq seed %.3Le   1 NR %.3Le   2 NR %.3Le\n", e[0], e[1], e[2]);
    printf("rcp seed %.3Le   1 NR %.3Le   2 NR %.3Le\n", e[3], e[4], e[5]);
    return 0;
}
