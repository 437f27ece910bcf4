// What does SQ_LDS_BANK_CONFLICT count for the access patterns of the solve kernel?  (89 % of the LDS-active
// cycles of the paper-horizon kernel are "bank conflict" cycles.)  One wavefront per workgroup, 4096 LDS reads per lane of
// each pattern; run under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE and
// compare the per-kernel ratios:
//   b32_linear     lane i reads dword i                      the conflict-free reference (32 banks x 4 B, 64 lanes: 2 passes)
//   b64_linear     lane i reads double i                     a full-rate 64-bit read (64 lanes x 8 B = 4 passes of 128 B)
//   b64_tile17     lane (g, j) reads double (4 ks + g) * 17 + j   the B-operand read of a 16 x 16 tile with row stride 17
//   b64_tile16     the same with row stride 16               (what the padding is there to avoid)
//   b64_crow17     lane (g, j) reads double g * 17 + j + 68 r     the accumulator-layout read / write of a tile
//   b64_bcast      every lane reads the same double          the wave-uniform broadcasts of the panel / chain code
//   b128_linear    lane i reads double2 i
// hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_conflict.hip -o gpurun_out/lds_conflict
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int REPS = 4096;

template <int PATTERN>
__global__ __launch_bounds__(64) void lds_pattern(double* out) {
    __shared__ __attribute__((aligned(16))) double s[2048];
    const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
    for (int i = lane; i < 2048; i += 64) s[i] = double(i);
    __syncthreads();
    double acc = 0.0;
    int base = 0;
    for (int r = 0; r < REPS; ++r) {
        asm volatile("" : "+v"(base));   // keep the read inside the loop
        if constexpr (PATTERN == 0) acc += double(reinterpret_cast<const float*>(s)[base + lane]);
        else if constexpr (PATTERN == 1) acc += s[base + lane];
        else if constexpr (PATTERN == 2) acc += s[base + (4 * (r & 3) + g) * 17 + j];
        else if constexpr (PATTERN == 3) acc += s[base + (4 * (r & 3) + g) * 16 + j];
        else if constexpr (PATTERN == 4) acc += s[base + g * 17 + j + 68 * (r & 3)];
        else if constexpr (PATTERN == 5) acc += s[base + (r & 255)];
        else { const double2 v = reinterpret_cast<const double2*>(s)[base + lane]; acc += v.x + v.y; }
    }
    out[blockIdx.x * 64 + lane] = acc;
}

int main() {
    double* d = nullptr;
    hipMalloc(&d, 256 * 64 * sizeof(double));
    hipLaunchKernelGGL(lds_pattern<0>, dim3(256), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(lds_pattern<1>, dim3(256), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(lds_pattern<2>, dim3(256), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(lds_pattern<3>, dim3(256), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(lds_pattern<4>, dim3(256), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(lds_pattern<5>, dim3(256), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(lds_pattern<6>, dim3(256), dim3(64), 0, 0, d);
    hipDeviceSynchronize();
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    hipFree(d);
    return 0;
}
