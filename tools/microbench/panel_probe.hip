// The 16-pivot panel stream of P3 in isolation: the C++ form the compiler schedules (panel_factor<1, 16> of
// vsmpc_kernels.hip, one row per lane, v_readlane broadcasts) against the hand-scheduled assembly of
// csrc/vsmpc_panel_asm.inc (tools/gen_panel_asm.py).  One wavefront per workgroup, one workgroup per CU; s_memtime around
// the stream, LDS loads and stores included in both.  Prints the median cycles of each and whether the results agree bit
// for bit.
//   hipcc --offload-arch=gfx950 -O3 -I paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd/csrc -o panel_probe panel_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
#define VS_DEV __device__ __forceinline__
VS_DEV double readlane_f64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
VS_DEV double fast_rsqrt(double d) {
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y, y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
}
#include "vsmpc_panel_asm.inc"

VS_DEV void panel16_cpp(const double* T, int lane, double (&a)[16], double& inv_mine, double& dmin, double& inv_last) {
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = T[c];
    dmin = 1.0; inv_mine = 1.0; inv_last = 1.0;
    double d = readlane_f64(a[0], 0);
    double inv = fast_rsqrt(d);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        dmin = fmin(dmin, d);
        inv_mine = lane == j ? inv : inv_mine;
        inv_last = inv;
        const double l = a[j] * inv;
        a[j] = l;
        if (j + 1 < 16) {
            const double lcj = readlane_f64(l, j + 1);
            a[j + 1] = fma(-l, lcj, a[j + 1]);
            d = readlane_f64(a[j + 1], j + 1);
            inv = fast_rsqrt(d);
        }
#pragma unroll
        for (int c = j + 2; c < 16; ++c) a[c] = fma(-l, readlane_f64(l, c), a[c]);
    }
}

// VARIANT 0: compiled C++ (lanes 0..15 the diagonal tile, lanes 16..63 the rows rowbase + lane), 1: DPP stream (every lane
// a panel row: 16 + lane; the diagonal tile in g)
template <int VARIANT>
__global__ __launch_bounds__(64) void probe(const double* __restrict__ in, double* __restrict__ out, unsigned long long* cyc, int rowbase) {
    __shared__ double sT[80 * 17 + 16];
    const int lane = threadIdx.x;
    for (int i = lane; i < 80 * 17; i += 64) sT[i] = in[i];
    __syncthreads();
    double a[16], inv_mine = 0, dmin, inv_last;
    const int row = VARIANT == 1 ? 16 + lane : (lane < 16 ? lane : rowbase + lane);
    double* T = sT + row * 17;
    const unsigned addr = unsigned(reinterpret_cast<uintptr_t>(T));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (VARIANT == 0) {
        panel16_cpp(T, lane, a, inv_mine, dmin, inv_last);
        if (lane >= 16) {
#pragma unroll
            for (int c = 0; c < 16; ++c) T[c] = a[c];
        } else {
            sT[80 * 17 + lane] = inv_mine;
        }
    } else {
        panel16x1_dpp(addr, addr, unsigned(reinterpret_cast<uintptr_t>(sT + (lane & 15) * 17)), unsigned(reinterpret_cast<uintptr_t>(sT + 80 * 17)), a, inv_last);
    }
    __syncthreads();
    if (lane < 16) {   // the factored diagonal tile goes back after the barrier, as in the solver
#pragma unroll
        for (int c = 0; c < 16; ++c) sT[lane * 17 + c] = a[c];
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0) {
        for (int i = lane; i < 80 * 17 + 16; i += 64) out[i] = sT[i];
        if (lane == 0) out[80 * 17 + 16] = inv_last;
    }
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    constexpr int N = 80 * 17, NO = N + 17, G = 256;
    std::vector<double> h(N);
    // rows 0..15: an SPD tile (lower triangle meaningful), rows 16..79: panel rows below it
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return double(s >> 8) / double(1u << 24) - 0.5; };
    for (int r = 0; r < 80; ++r)
        for (int c = 0; c < 17; ++c) h[r * 17 + c] = rnd();
    for (int r = 0; r < 16; ++r) {
        for (int c = 0; c < 16; ++c) h[r * 17 + c] = h[std::max(r, c) * 17 + std::min(r, c)] * 0.2;
        h[r * 17 + r] = 3.0 + 0.1 * r;
    }
    double *din, *dout;
    unsigned long long* dcyc;
    (void)hipMalloc(&din, N * 8); (void)hipMalloc(&dout, NO * 8); (void)hipMalloc(&dcyc, G * 8);
    (void)hipMemcpy(din, h.data(), N * 8, hipMemcpyHostToDevice);
    std::vector<double> ref(NO), o(NO);
    const char* names[2] = {"compiled C++ stream, v_readlane broadcasts (48 rows + tile)", "DPP row_newbcast stream in assembly (64 rows + tile)      "};
    int bad = 0;
    for (int v = 0; v < 2; ++v) {
        std::vector<unsigned long long> med;
        for (int rep = 0; rep < 5; ++rep) {
            if (v == 0) hipLaunchKernelGGL(probe<0>, dim3(G), dim3(64), 0, 0, din, dout, dcyc, 0);
            else hipLaunchKernelGGL(probe<1>, dim3(G), dim3(64), 0, 0, din, dout, dcyc, 0);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            std::vector<unsigned long long> c(G);
            (void)hipMemcpy(c.data(), dcyc, G * 8, hipMemcpyDeviceToHost);
            std::sort(c.begin(), c.end());
            med.push_back(c[G / 2]);
        }
        std::sort(med.begin(), med.end());
        (void)hipMemcpy(o.data(), dout, NO * 8, hipMemcpyDeviceToHost);
        int diff = 0;
        if (v == 0) {   // reference = the C++ stream on rows 16..63 and (second launch) 32..79
            ref = o;
            hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, din, dout, dcyc, 16);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(o.data(), dout, NO * 8, hipMemcpyDeviceToHost);
            for (int r = 64; r < 80; ++r)
                for (int c = 0; c < 17; ++c) ref[r * 17 + c] = o[r * 17 + c];
        } else {
            for (int i = 0; i < NO; ++i) {
                if (i < N && i % 17 == 16) continue;
                if (i < 16 * 17 && i % 17 > i / 17) continue;      // above the diagonal: leftovers
                diff += std::memcmp(&ref[i], &o[i], 8) != 0;
            }
        }
        printf("%s: median %llu cycles (runs %llu .. %llu)%s\n", names[v], med[2], med[0], med[4],
               v == 0 ? "" : diff ? "  DIFFERS from the C++ stream" : "  bit-identical to the C++ stream (L, 1/L_jj)");
        bad += diff;
    }
    printf("L[1][0] = %.17g, 1/L_00 = %.17g, 1/L_15,15 = %.17g\n", ref[17], ref[N], ref[N + 16]);
    return bad != 0;
}
