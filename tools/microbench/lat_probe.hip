// Instruction cost probe for the serial parts of the solve kernel (gfx950): one wavefront, s_memtime around unrolled
// sequences.  Build: hipcc --offload-arch=gfx950 -O3 -o lat_probe lat_probe.hip ; run on the GPU box.
// Prints cycles per instruction (s_memtime ticks at the shader clock... s_memtime counts at a fixed 100 MHz on some parts:
// the probe also times 1000 dependent v_add_u32 to calibrate).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define REP256(x) REP64(x) REP64(x) REP64(x) REP64(x)
__device__ __forceinline__ double readlane_f64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
__global__ void probe(unsigned long long* out, double* sink, double seed) {
    double a = seed + threadIdx.x, b = seed * 0.5, c = 1.0 + seed;
    double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
    unsigned long long t0, t1;
    int k = 0;
    unsigned u = threadIdx.x;
    // 0: dependent v_add_u32 chain (calibration: 1 VALU issue = 4 cycles at best)
    t0 = __builtin_amdgcn_s_memtime();
    REP256(asm volatile("v_add_u32 %0, %0, 1" : "+v"(u));)
    t1 = __builtin_amdgcn_s_memtime(); out[k++] = t1 - t0;
    // 1: dependent f64 fma chain
    t0 = __builtin_amdgcn_s_memtime();
    REP256(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
    t1 = __builtin_amdgcn_s_memtime(); out[k++] = t1 - t0;
    // 2: eight independent f64 fma chains (throughput)
    t0 = __builtin_amdgcn_s_memtime();
    REP64(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                       "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                       : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(c));)
    t1 = __builtin_amdgcn_s_memtime(); out[k++] = t1 - t0;   // 512 instructions
    // 3: independent v_readlane_b32 (throughput)
    int s0, s1;
    t0 = __builtin_amdgcn_s_memtime();
    REP256(asm volatile("v_readlane_b32 %0, %2, 3\n v_readlane_b32 %1, %3, 5" : "=s"(s0), "=s"(s1) : "v"(u), "v"(k));)
    t1 = __builtin_amdgcn_s_memtime(); out[k++] = t1 - t0;   // 512 instructions
    // 4: readlane pair -> fma with the scalar pair (the broadcast-update pattern), independent accumulators
    t0 = __builtin_amdgcn_s_memtime();
    REP64(asm volatile("v_readlane_b32 s20, %4, 3\n v_readlane_b32 s21, %5, 3\n v_fma_f64 %0, %6, s[20:21], %0\n"
                       "v_readlane_b32 s22, %4, 4\n v_readlane_b32 s23, %5, 4\n v_fma_f64 %1, %6, s[22:23], %1\n"
                       "v_readlane_b32 s24, %4, 5\n v_readlane_b32 s25, %5, 5\n v_fma_f64 %2, %6, s[24:25], %2\n"
                       "v_readlane_b32 s26, %4, 6\n v_readlane_b32 s27, %5, 6\n v_fma_f64 %3, %6, s[26:27], %3"
                       : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(u), "v"(k), "v"(b)
                       : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
    t1 = __builtin_amdgcn_s_memtime(); out[k++] = t1 - t0;   // 768 instructions (256 broadcasts + updates)
    // 5: the pivot chain: mul -> readlane pair -> fma -> readlane pair -> rsq -> 5 dependent ops (compiler-scheduled)
    {
        double inv = x6, aj = a;
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const double l = aj * inv;
            const double sc = readlane_f64(l, 3);
            const double a2 = fma(-l, sc, c);
            const double d = readlane_f64(a2, 4);
            const double y = __builtin_amdgcn_rsq(d);
            const double e = fma(-d * y, y, 1.0);
            inv = fma(y * e, fma(0.375, e, 0.5), y);
            aj = a2;
        }
        t1 = __builtin_amdgcn_s_memtime(); out[k++] = t1 - t0;   // 64 pivots
        a += inv + aj;
    }
    // 6: dependent v_rsq_f64
    t0 = __builtin_amdgcn_s_memtime();
    REP256(asm volatile("v_rsq_f64 %0, %0" : "+v"(x7));)
    t1 = __builtin_amdgcn_s_memtime(); out[k++] = t1 - t0;
    // 7: dependent mfma f64 16x16x4 chain
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 acc = {a, b, c, a};
    t0 = __builtin_amdgcn_s_memtime();
    REP64(acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0);)
    asm volatile("s_nop 7\n s_nop 7\n s_nop 7" ::: );
    t1 = __builtin_amdgcn_s_memtime(); out[k++] = t1 - t0;   // 64
    // 8: mfma whose B operand is the previous result's register 0 (the trsm chain)
    t0 = __builtin_amdgcn_s_memtime();
    REP64(acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, acc[0], acc, 0, 0, 0);)
    asm volatile("s_nop 7\n s_nop 7\n s_nop 7" ::: );
    t1 = __builtin_amdgcn_s_memtime(); out[k++] = t1 - t0;   // 64
    // 9: LDS round trip: ds_write then dependent uniform ds_read, 64 times
    __shared__ double sh[256];
    double v = a;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 64; ++i) { sh[threadIdx.x] = v; v = sh[(i * 7) & 63] + 1.0; }
    t1 = __builtin_amdgcn_s_memtime(); out[k++] = t1 - t0;   // 64 round trips
    // 10: 64 ds_write_b64 of all lanes to ONE address / 11: to per-lane addresses / 12: b128 one address / 13: b128 per lane
    {
        __shared__ double sw[1024];
        double* p1 = sw + (seed > 100.0 ? 1 : 0);
        double* p2 = sw + threadIdx.x;
        double* p4 = sw + 2 * threadIdx.x;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 64; ++i) { p1[(i & 7) * 2] = v; asm volatile("" ::: "memory"); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime(); out[k++] = t1 - t0;
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 64; ++i) { p2[(i & 7) * 64] = v; asm volatile("" ::: "memory"); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime(); out[k++] = t1 - t0;
        double2 vv = make_double2(v, v + 1);
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 64; ++i) { reinterpret_cast<double2*>(p1)[(i & 7)] = vv; asm volatile("" ::: "memory"); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime(); out[k++] = t1 - t0;
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 64; ++i) { reinterpret_cast<double2*>(p4)[(i & 3) * 64] = vv; asm volatile("" ::: "memory"); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime(); out[k++] = t1 - t0;
        v += sw[threadIdx.x];
    }
    // 14: dependent mfma through C, pinned with asm
    {
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 c4 = {a, b, c, a};
        t0 = __builtin_amdgcn_s_memtime();
        REP64(asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c4) : "v"(b), "v"(c));)
        asm volatile("s_nop 7\n s_nop 7\n s_nop 7" ::: );
        t1 = __builtin_amdgcn_s_memtime(); out[k++] = t1 - t0;
        // 15: mfma -> VALU read of the result -> mfma (B operand), the trsm / diag chain
        double bb = b;
        t0 = __builtin_amdgcn_s_memtime();
        REP64(c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, bb, c4, 0, 0, 0); asm volatile("" : "+v"(c4)); bb = c4[0] + b; asm volatile("" : "+v"(bb));)
        t1 = __builtin_amdgcn_s_memtime(); out[k++] = t1 - t0;
        a += c4[0] + c4[1] + c4[2] + c4[3] + bb;
    }
    // 16: permlane16_swap + permlane32_swap dependent pairs x64
    {
        unsigned p = u, q = u + 1;
        t0 = __builtin_amdgcn_s_memtime();
        REP64(asm volatile("v_permlane16_swap_b32 %0, %1\n s_nop 1\n v_permlane32_swap_b32 %0, %1\n s_nop 1" : "+v"(p), "+v"(q));)
        t1 = __builtin_amdgcn_s_memtime(); out[k++] = t1 - t0;
        u += p + q;
    }
    // 17: readlane -> dependent VALU -> readlane chain x64 (latency of the broadcast path)
    {
        unsigned p = u;
        int sr;
        t0 = __builtin_amdgcn_s_memtime();
        REP64(asm volatile("v_readlane_b32 %1, %0, 3\n s_nop 3\n v_add_u32 %0, %0, %1" : "+v"(p), "=s"(sr));)
        t1 = __builtin_amdgcn_s_memtime(); out[k++] = t1 - t0;
        u += p;
    }
    // 10: s_memrealtime vs s_memtime over the whole probe
    out[k++] = __builtin_amdgcn_s_memrealtime();
    sink[threadIdx.x] = a + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + u + s0 + s1 + acc[0] + acc[1] + acc[2] + acc[3] + v;
}
int main() {
    unsigned long long* out; double* sink;
    if (hipMalloc(&out, 64 * 8) != hipSuccess || hipMalloc(&sink, 64 * 8) != hipSuccess) return 1;
    const char* names[] = {"dependent v_add_u32 x256", "dependent v_fma_f64 x256", "8 independent fma chains x512", "independent v_readlane_b32 x512",
                           "readlane pair + fma (x256 updates, 768 instr)", "pivot chain x64", "dependent v_rsq_f64 x256", "dependent mfma (C) x64",
                           "mfma via B operand x64", "LDS write->read round trip x64", "(realtime)", "ds_write_b64 one address x64", "ds_write_b64 per-lane x64", "ds_write_b128 one address x64", "ds_write_b128 per-lane x64", "dependent mfma asm x64", "mfma->valu->mfma x64", "permlane16+32 swap pair x64", "readlane->valu chain x64"};
    const int div[] = {256, 256, 512, 512, 256, 64, 256, 64, 64, 64, 1, 64, 64, 64, 64, 64, 64, 64, 64};
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, sink, 1.25);
        if (hipDeviceSynchronize() != hipSuccess) return 2;
    }
    unsigned long long h[32];
    if (hipMemcpy(h, out, 19 * 8, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    for (int i = 0; i < 19; ++i) printf("%-48s %8llu ticks  %.1f per unit\n", names[i], h[i], double(h[i]) / div[i]);
    return 0;
}
