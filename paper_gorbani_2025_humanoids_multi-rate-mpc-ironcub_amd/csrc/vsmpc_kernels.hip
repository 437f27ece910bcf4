// Hand-written HIP kernels for gfx950 (MI355X): one workgroup per MPC instance.
//
// Replaces, per instance, IMPCProblem::update + IMPCProblem::solve + VariableSamplingMPC::solveMPC
// (IMPCProblem.cpp:150-298, variableSamplingMPC.cpp:88-112) with a structure-exploiting exact solve:
//
//   P0 linearise   A, Bj, Bt, c in LDS                       (systemDynamicsVSMPC.cpp:79-103,288-319,384-429)
//   P1 condense    sensitivity recursion, one condensed column per thread in registers;
//                  C = sum_k Y_k^T Y_k with v_mfma_f64_16x16x4_f64, accumulators in registers
//                                                            (constraintsVSMPC.cpp:76-131, costsVSMPC.cpp:166-178)
//   P2 augment     M = C + R, gradient row                   (costsVSMPC.cpp:375-409,468-487,558-592)
//   P3 cholesky    blocked right-looking LL^T on 16x16 LDS tiles, trailing updates on MFMA
//   P4 box QP      Schur complement on the warped throttles, block principal pivoting, one wavefront
//                                                            (constraintsVSMPC.cpp:338-365)
//   P5 back-subst  joints from the factor
//   P6 simulate    state trajectory, primal in the reference variable order, first-move block
//                                                            (variableSamplingMPC.cpp:93-108,138-151)
//
// FP64 throughout.  The un-condensed KKT system the reference hands to OSQP has condition number
// ~1e12; the condensed Hessian factored here is benign (1e2..1e3).
#include "vsmpc_device.hpp"
#include "vsmpc_launch.hpp"

namespace vsmpc {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int BLOCK = 256;
constexpr int NWAVES = BLOCK / 64;

VS_DEV double readlane_f64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------------------------------------
// LDS carve-up (doubles)
// ------------------------------------------------------------------------------------------------
template <class D>
struct Smem {
    static constexpr int oIn = 0;
    static constexpr int oA = oIn + ((D::NIN + 3) & ~3);
    static constexpr int oBj = oA + NX * NX;
    static constexpr int oBt = oBj + NX * NJ;
    static constexpr int oC = oBt + NX * NTH;
    static constexpr int oVprev = oC + 28;
    static constexpr int oInvD = oVprev + 4;
    static constexpr int oW = oInvD + D::NP;
    static constexpr int oZ = oW + D::NP;
    static constexpr int oSv = oZ + D::NP;
    static constexpr int oSvec = oSv + D::NV * (D::NV + 1);
    static constexpr int oV = oSvec + D::NV;
    static constexpr int oX = oV + D::NV;
    static constexpr int oFlags = oX + D::NXS;       // 4 doubles worth of int flags
    static constexpr int oY = (oFlags + 4 + 3) & ~3;
    static constexpr int oM = oY + 2 * 20 * D::YS;
    static constexpr int total = oM + D::NTRI * D::TS;
    static constexpr size_t bytes = size_t(total) * sizeof(double);
};

template <class D>
VS_DEV int tile_off(int i, int j) { return (i * (i + 1) / 2 + j) * D::TS; }

// element (gr, gc), gc <= gr, of the lower-triangular tile storage
template <class D>
VS_DEV int lower_at(int gr, int gc) {
    return tile_off<D>(gr >> 4, gc >> 4) + (gr & 15) * 17 + (gc & 15);
}

// ------------------------------------------------------------------------------------------------
// P0: linearisation into LDS (dense, row-major) — also the body of the linearise-only kernel
// ------------------------------------------------------------------------------------------------
template <class D>
VS_DEV void p0_linearize(const DevCfg& cfg, const double* __restrict__ sIn, double* __restrict__ sA,
                         double* __restrict__ sBj, double* __restrict__ sBt, double* __restrict__ sC,
                         double* __restrict__ sVprev, int tid, int nthreads) {
    for (int i = tid; i < NX * NX + NX * NJ + NX * NTH + 28; i += nthreads) sA[i] = 0.0;  // A,Bj,Bt,c contiguous
    __syncthreads();
    if (tid == 0) {
        // A[rpy, angMom] = W(rpy)^-1 * I_G^-1                       (systemDynamicsVSMPC.cpp:86-87,140-147)
        const double* I = sIn + VSMPC_IN_INERTIA;
        const double a = I[0], b = I[1], c = I[2], d = I[3], e = I[4], f = I[5], g = I[6], h = I[7], k = I[8];
        const double A00 = e * k - f * h, A01 = c * h - b * k, A02 = b * f - c * e;
        const double A10 = f * g - d * k, A11 = a * k - c * g, A12 = c * d - a * f;
        const double A20 = d * h - e * g, A21 = b * g - a * h, A22 = a * e - b * d;
        const double idet = 1.0 / (a * A00 + b * A10 + c * A20);
        const double Ii[9] = {A00 * idet, A01 * idet, A02 * idet, A10 * idet, A11 * idet,
                              A12 * idet, A20 * idet, A21 * idet, A22 * idet};
        const double r = sIn[VSMPC_IN_RPY + 0], p = sIn[VSMPC_IN_RPY + 1];
        const double sr = sin(r), cr = cos(r), cp = cos(p), tp = tan(p);
        const double Wi[9] = {1.0, sr * tp, cr * tp, 0.0, cr, -sr, 0.0, sr / cp, cr / cp};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0.0;
                for (int q = 0; q < 3; ++q) s += Wi[3 * i + q] * Ii[3 * q + j];
                sA[(6 + i) * NX + 9 + j] = s;
            }
    } else if (tid >= 1 && tid <= 4) {
        // jets                                                      (systemDynamicsVSMPC.cpp:384-429)
        const int i = tid - 1;
        sVprev[i] = Jet::v_of_throttle(sIn[VSMPC_IN_UPREV + i]);
        if (cfg.use_jet) {
            const double T0 = sIn[VSMPC_IN_T0 + i], Td0 = sIn[VSMPC_IN_TD0 + i], up = sIn[VSMPC_IN_UPREV + i];
            const double dhT = Jet::dh_dT(T0, Td0, up), dhTd = Jet::dh_dTd(T0, Td0, up);
            sA[(12 + i) * NX + 16 + i] = 1.0;
            sA[(16 + i) * NX + 12 + i] = dhT;
            sA[(16 + i) * NX + 16 + i] = dhTd;
            sBt[(16 + i) * NTH + i] = Jet::G(sIn[VSMPC_IN_TDES + i], sIn[VSMPC_IN_TDDES + i]);
            sC[16 + i] = Jet::F(T0, Td0) - dhT * T0 - dhTd * Td0;
        } else {
            sBt[(12 + i) * NTH + i] = 1.0;
        }
    } else if (tid == 5) {
        // CoM kinematics, -S(omega) blocks, gravity term, integrators  (systemDynamicsVSMPC.cpp:90-91,296-316)
        const double m = sIn[VSMPC_IN_MASS], im = 1.0 / m;
        const double* R = sIn + VSMPC_IN_WRB;
        const double* w = sIn + VSMPC_IN_OMEGA;
        const double* gr = sIn + VSMPC_IN_GRAV;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) sA[i * NX + 3 + j] = im * R[3 * i + j];
        const double S[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};  // FlightControlUtils.cpp:77-85
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                sA[(3 + i) * NX + 3 + j] = -S[3 * i + j];
                sA[(9 + i) * NX + 9 + j] = -S[3 * i + j];
            }
        const double am = sIn[VSMPC_IN_ALPHA] * m;
        for (int i = 0; i < 3; ++i) {
            sC[3 + i] = am * (R[0 + i] * gr[0] + R[3 + i] * gr[1] + R[6 + i] * gr[2]);  // alpha*m*R^T g
            sA[(20 + i) * NX + i] = 1.0;
            sA[(23 + i) * NX + 6 + i] = 1.0;
            sC[20 + i] = -sIn[VSMPC_IN_PREF + i];
            sC[23 + i] = -sIn[VSMPC_IN_RPYINIT + i];
        }
    } else if (tid >= 8 && tid < 32) {
        // thrust maps A[linMom|angMom, T] = A_mom,body                (systemDynamicsVSMPC.cpp:92-93,303-304)
        const int e = tid - 8, r = e >> 2, j = e & 3;  // r in 0..5
        const int row = r < 3 ? 3 + r : 6 + r;         // 3..5, 9..11
        sA[row * NX + 12 + j] = sIn[VSMPC_IN_AMOM + e];
    } else if (tid >= 32 && tid < 56) {
        const int e = tid - 32, r = e >> 3, j = e & 7;  // Lambda_lin,B -> Bj[3..5]   (:305-306)
        sBj[(3 + r) * NJ + j] = sIn[VSMPC_IN_LLIN + e];
    } else if (tid >= 56 && tid < 80) {
        const int e = tid - 56, r = e >> 3, j = e & 7;  // Lambda_ang,B -> Bj[9..11]  (:94-95)
        sBj[(9 + r) * NJ + j] = sIn[VSMPC_IN_LANG + e];
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// linearise-only kernel (vsmpc_linearize_batch)
// ------------------------------------------------------------------------------------------------
template <class D>
__global__ __launch_bounds__(128) void linearize_kernel(DevCfg cfg, const double* __restrict__ in,
                                                         double* __restrict__ A, double* __restrict__ Bj,
                                                         double* __restrict__ Bt, double* __restrict__ c) {
    __shared__ double sIn[(D::NIN + 3) & ~3];
    __shared__ double sLin[NX * NX + NX * NJ + NX * NTH + 28 + 4];
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int i = tid; i < D::NIN; i += 128) sIn[i] = in[size_t(b) * D::NIN + i];
    __syncthreads();
    double* sA = sLin;
    double* sBj = sA + NX * NX;
    double* sBt = sBj + NX * NJ;
    double* sC = sBt + NX * NTH;
    double* sVprev = sC + 28;
    p0_linearize<D>(cfg, sIn, sA, sBj, sBt, sC, sVprev, tid, 128);
    for (int i = tid; i < NX * NX; i += 128) A[size_t(b) * NX * NX + i] = sA[i];
    for (int i = tid; i < NX * NJ; i += 128) Bj[size_t(b) * NX * NJ + i] = sBj[i];
    for (int i = tid; i < NX * NTH; i += 128) Bt[size_t(b) * NX * NTH + i] = sBt[i];
    for (int i = tid; i < NX; i += 128) c[size_t(b) * NX + i] = sC[i];
}

// ------------------------------------------------------------------------------------------------
// cost terms on the condensed inputs (P2)
// ------------------------------------------------------------------------------------------------
template <class D>
VS_DEV double input_cost_term(const DevCfg& cfg, const double* __restrict__ sIn,
                              const double* __restrict__ sVprev, int gr, int gc) {
    if (gr < D::NU) return (gr == gc) ? cfg.wj[gr & 7] : 0.0;  // (65000+20) I  (costsVSMPC.cpp:375-381,564-571)
    if (gr < D::NZ) {
        if (gc < D::NU) return 0.0;
        const int q1 = gr - D::NU, q2 = gc - D::NU;
        if ((q1 & 3) != (q2 & 3)) return 0.0;
        const int b1 = v_block_of_internal<D>(q1), b2 = v_block_of_internal<D>(q2);
        if (b1 == b2)  // first-difference penalty + v0 anchor  (costsVSMPC.cpp:383-409,472-476)
            return cfg.w_thr * double((b1 > 0) + (b1 < D::NVB - 1)) + (b1 == 0 ? cfg.w_init : 0.0);
        const int db = b1 - b2;
        return (db == 1 || db == -1) ? -cfg.w_thr : 0.0;
    }
    if (gr == D::NZ && gc < D::NZ) {  // gradient row
        if (gc < D::NU) return cfg.w_reg * sIn[VSMPC_IN_QERR + (gc & 7)];      // costsVSMPC.cpp:586-590
        const int q = gc - D::NU;
        return v_block_of_internal<D>(q) == 0 ? -cfg.w_init * sVprev[q & 3] : 0.0;  // costsVSMPC.cpp:479-485
    }
    return 0.0;
}

// ------------------------------------------------------------------------------------------------
// the solve kernel
// ------------------------------------------------------------------------------------------------
template <class D>
__global__ __launch_bounds__(BLOCK, 1) void solve_kernel(DevCfg cfg, const double* __restrict__ in, int batch,
                                                         double* __restrict__ xout, double* __restrict__ fmout,
                                                         int* __restrict__ status_out, int* __restrict__ iters_out,
                                                         double* __restrict__ dbgM, double* __restrict__ dbgL) {
    using S = Smem<D>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sIn = smem + S::oIn;
    double* sA = smem + S::oA;
    double* sBj = smem + S::oBj;
    double* sBt = smem + S::oBt;
    double* sC = smem + S::oC;
    double* sVprev = smem + S::oVprev;
    double* sInvD = smem + S::oInvD;
    double* sW = smem + S::oW;
    double* sZ = smem + S::oZ;
    double* sSv = smem + S::oSv;
    double* sSvec = smem + S::oSvec;
    double* sV = smem + S::oV;
    double* sX = smem + S::oX;
    int* sFlags = reinterpret_cast<int*>(smem + S::oFlags);  // [0]=numerical failure, [1]=status, [2]=iters
    double* sY = smem + S::oY;
    double* sM = smem + S::oM;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int inst = blockIdx.x;
    if (inst >= batch) return;

    // ---------------------------------------------------------------- P0
    for (int i = tid; i < D::NIN; i += BLOCK) sIn[i] = in[size_t(inst) * D::NIN + i];
    for (int i = tid; i < 2 * 20 * D::YS; i += BLOCK) sY[i] = 0.0;  // rows 18,19 and pad columns stay zero
    if (tid < 4) sFlags[tid] = 0;
    __syncthreads();
    p0_linearize<D>(cfg, sIn, sA, sBj, sBt, sC, sVprev, tid, BLOCK);

    // ---------------------------------------------------------------- P1 condense
    // tiles of the lower triangle are dealt round-robin to the four wavefronts
    constexpr int TPW = (D::NTRI + NWAVES - 1) / NWAVES;
    d4 acc[TPW];
    int ti[TPW], tj[TPW], tstart[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        acc[q] = d4{0.0, 0.0, 0.0, 0.0};
        const int t = q * NWAVES + wave;
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= t) ++i;
        ti[q] = i;
        tj[q] = t - i * (i + 1) / 2;
        if (t < D::NTRI) {
            const int a = tile_first_stage<D>(ti[q]), b = tile_first_stage<D>(tj[q]);
            tstart[q] = a > b ? a : b;
        } else {
            ti[q] = tj[q] = 0;
            tstart[q] = 1 << 20;
        }
    }

    // per-column state of the sensitivity recursion (threads 0..NP-1 own one condensed column each)
    const int col = tid;
    int kind = 3, blk = 0, comp = 0;  // 0 = joint column, 1 = throttle column, 2 = affine column, 3 = pad
    if (col < D::NU) { kind = 0; blk = col >> 3; comp = col & 7; }
    else if (col < D::NZ) { kind = 1; blk = v_block_of_internal<D>(col - D::NU); comp = (col - D::NU) & 3; }
    else if (col == D::NZ) { kind = 2; }

    double s[NX];
    double bl[3], ba[3], bT[4], bTd[4];
#pragma unroll
    for (int r = 0; r < NX; ++r) s[r] = (kind == 2) ? sIn[VSMPC_IN_X0 + r] : 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        bl[r] = kind == 0 ? sBj[(3 + r) * NJ + comp] : (kind == 2 ? sC[3 + r] : 0.0);
        ba[r] = kind == 0 ? sBj[(9 + r) * NJ + comp] : (kind == 2 ? sC[9 + r] : 0.0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bT[i] = kind == 1 ? (i == comp ? sBt[(12 + i) * NTH + i] : 0.0) : (kind == 2 ? sC[12 + i] : 0.0);
        bTd[i] = kind == 1 ? (i == comp ? sBt[(16 + i) * NTH + i] : 0.0) : (kind == 2 ? sC[16 + i] : 0.0);
    }

    for (int k = 0; k < D::N; ++k) {
        double* Yb = sY + (k & 1) * 20 * D::YS;
        if (tid < D::NP) {
            const double dt = cfg.dt[k];
            const bool actJ = (kind == 0 && joint_block_of_stage<D>(k) == blk) || kind == 2;
            const bool actT = (kind == 1 && throttle_block_of_stage<D>(k) == blk) || kind == 2;
            double ds[NX];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                double a0 = 0.0, a1 = actJ ? bl[r] : 0.0, a2 = 0.0, a3 = actJ ? ba[r] : 0.0;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    a0 += sA[r * NX + 3 + j] * s[3 + j];
                    a1 += sA[(3 + r) * NX + 3 + j] * s[3 + j];
                    a2 += sA[(6 + r) * NX + 9 + j] * s[9 + j];
                    a3 += sA[(9 + r) * NX + 9 + j] * s[9 + j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a1 += sA[(3 + r) * NX + 12 + j] * s[12 + j];
                    a3 += sA[(9 + r) * NX + 12 + j] * s[12 + j];
                }
                ds[r] = a0;
                ds[3 + r] = a1;
                ds[6 + r] = a2;
                ds[9 + r] = a3;
                ds[20 + r] = s[r] + (kind == 2 ? sC[20 + r] : 0.0);
                ds[23 + r] = s[6 + r] + (kind == 2 ? sC[23 + r] : 0.0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ds[12 + i] = sA[(12 + i) * NX + 16 + i] * s[16 + i] + (actT ? bT[i] : 0.0);
                ds[16 + i] = sA[(16 + i) * NX + 12 + i] * s[12 + i] + sA[(16 + i) * NX + 16 + i] * s[16 + i]
                             + (actT ? bTd[i] : 0.0);
            }
#pragma unroll
            for (int r = 0; r < NX; ++r) s[r] += dt * ds[r];
            // Y_k = sqrt(Q) (S_k - xref_k on the affine column); reference column map costsVSMPC.cpp:191-200
            const int rc = k < D::NS ? 0 : k - D::NS;
#pragma unroll
            for (int r = 0; r < NWROWS; ++r) {
                double v = s[wrow(r)];
                if (kind == 2 && r < 12) v -= sIn[VSMPC_IN_XREF + rc * 12 + r];
                Yb[r * D::YS + col] = cfg.sq[r] * v;
            }
        }
        __syncthreads();
        // C += Y_k^T Y_k : D = A*B with A[m][kk] = Y[kk][16 i + m], B[kk][n] = Y[kk][16 j + n]
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            const double* yrow = Yb + (4 * ks + (lane >> 4)) * D::YS + (lane & 15);
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                if (k >= tstart[q]) {
                    const double a = yrow[16 * ti[q]];
                    const double b = yrow[16 * tj[q]];
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
                }
            }
        }
        // one barrier per node is enough: the next node writes the other Y buffer
    }
    __syncthreads();

    // ---------------------------------------------------------------- P2 augment: M = C + R, gradient row
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int t = q * NWAVES + wave;
        if (t < D::NTRI) {
            double* T = sM + tile_off<D>(ti[q], tj[q]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (lane >> 4) + 4 * r, c = lane & 15;
                const int gr = 16 * ti[q] + row, gc = 16 * tj[q] + c;
                double v = acc[q][r];
                // the gradient row also picks up the transposed element for the straddling diagonal tile
                v += input_cost_term<D>(cfg, sIn, sVprev, gr, gc);
                T[row * 17 + c] = v;
            }
        }
    }
    __syncthreads();
    if (dbgM != nullptr) {
        for (int e = tid; e < D::NP * D::NP; e += BLOCK) {
            const int gr = e / D::NP, gc = e % D::NP;
            dbgM[size_t(inst) * D::NP * D::NP + e] = gc <= gr ? sM[lower_at<D>(gr, gc)] : 0.0;
        }
        __syncthreads();
    }

    // ---------------------------------------------------------------- P3 blocked Cholesky (first NZ pivots)
    for (int p = 0; p < D::NT; ++p) {
        const int npiv = (D::NZ - 16 * p) < 16 ? (D::NZ - 16 * p) : 16;
        double* Tpp = sM + tile_off<D>(p, p);
        if (wave == 0) {
            // lane r (mod 16) owns row r of the diagonal tile; pivots broadcast with v_readlane
            const int r = lane & 15;
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = Tpp[r * 17 + c];
            int bad = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < npiv) {
                    const double d = readlane_f64(a[j], j);
                    bad |= !(d > 0.0);
                    const double inv = rsqrt(d);
                    const double lj = a[j] * inv;
                    a[j] = lj;
                    if (lane == 0) sInvD[16 * p + j] = inv;
#pragma unroll
                    for (int c = j + 1; c < 16; ++c) {
                        const double lcj = readlane_f64(lj, c);
                        a[c] -= lj * lcj;
                    }
                }
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) Tpp[r * 17 + c] = (c <= r) ? a[c] : 0.0;
            }
            if (bad && lane == 0) sFlags[0] = 1;
        }
        __syncthreads();
        if (p + 1 < D::NT) {
            // panel solve: one thread per row below the diagonal tile, X L_pp^T = A
            const int nrows = 16 * (D::NT - 1 - p);
            if (tid < nrows) {
                double* T = sM + tile_off<D>(p + 1 + (tid >> 4), p) + (tid & 15) * 17;
                double x[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    double v = T[j];
#pragma unroll
                    for (int c = 0; c < j; ++c) v -= x[c] * Tpp[j * 17 + c];
                    x[j] = v * sInvD[16 * p + j];
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) T[j] = x[j];
            }
            __syncthreads();
            // trailing update M_ij -= L_ip L_jp^T, p < j <= i, on the matrix cores
            const int m = D::NT - 1 - p;
            const int npairs = m * (m + 1) / 2;
            for (int q = wave; q < npairs; q += NWAVES) {
                int ii = 0;
                while ((ii + 1) * (ii + 2) / 2 <= q) ++ii;
                const int jj = q - ii * (ii + 1) / 2;
                const int i = p + 1 + ii, j = p + 1 + jj;
                double* Tij = sM + tile_off<D>(i, j);
                const double* Lip = sM + tile_off<D>(i, p) + (lane & 15) * 17 + (lane >> 4);
                const double* Ljp = sM + tile_off<D>(j, p) + (lane & 15) * 17 + (lane >> 4);
                d4 c4;
#pragma unroll
                for (int r = 0; r < 4; ++r) c4[r] = Tij[((lane >> 4) + 4 * r) * 17 + (lane & 15)];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lip[4 * ks], Ljp[4 * ks], c4, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) Tij[((lane >> 4) + 4 * r) * 17 + (lane & 15)] = c4[r];
            }
            __syncthreads();
        }
    }
    if (dbgL != nullptr) {
        for (int e = tid; e < D::NP * D::NP; e += BLOCK) {
            const int gr = e / D::NP, gc = e % D::NP;
            dbgL[size_t(inst) * D::NP * D::NP + e] = gc <= gr ? sM[lower_at<D>(gr, gc)] : 0.0;
        }
        __syncthreads();
    }

    // ---------------------------------------------------------------- P4 box QP on the throttles
    // Schur complement S = L22 L22^T, s = L22 (L^-1 g)_v   (row NZ of the factor holds L^-1 g)
    for (int e = tid; e < D::NV * D::NV; e += BLOCK) {
        const int r = e / D::NV, c = e % D::NV;
        const int kmax = r < c ? r : c;
        double sum = 0.0;
        for (int k = 0; k <= kmax; ++k)
            sum += sM[lower_at<D>(D::NU + r, D::NU + k)] * sM[lower_at<D>(D::NU + c, D::NU + k)];
        sSv[r * (D::NV + 1) + c] = sum;
    }
    if (tid < D::NV) {
        double sum = 0.0;
        for (int k = 0; k <= tid; ++k) sum += sM[lower_at<D>(D::NU + tid, D::NU + k)] * sM[lower_at<D>(D::NZ, D::NU + k)];
        sSvec[tid] = sum;
    }
    if (tid < D::NP) sW[tid] = tid < D::NZ ? -sM[lower_at<D>(D::NZ, tid)] : 0.0;  // y = -L^-1 g
    __syncthreads();

    if (wave == 0) {
        const int r = lane < D::NV ? lane : D::NV - 1;  // lanes >= NV shadow the last row (results unused)
        const bool valid = lane < D::NV;
        double row[D::NV];
#pragma unroll
        for (int c = 0; c < D::NV; ++c) row[c] = sSv[r * (D::NV + 1) + c];
        const double svr = sSvec[r];
        const bool hold = sIn[VSMPC_IN_HOLD] != 0.0;
        const bool fixed = valid && hold && (r >= D::NV - 4);  // v0 is the trailing block
        const double lo = fixed ? sVprev[r & 3] : cfg.vmin;    // constraintsVSMPC.cpp:351-364
        const double hi = fixed ? sVprev[r & 3] : cfg.vmax;
        int state = fixed ? -1 : 0;  // 0 free, -1 at lower, +1 at upper
        double v = 0.0;
        double gmax = fabs(svr);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gmax = fmax(gmax, __shfl_xor(gmax, o));
        const double gtol = 1e-10 * (1.0 + gmax);
        int best = D::NV + 1, patience = 3, status = VSMPC_STATUS_MAX_ITER, iters = 0;
        for (int it = 0; it < cfg.max_as_iter; ++it) {
            iters = it + 1;
            const bool isF = valid && state == 0;
            const unsigned long long Fmask = __ballot(isF);
            const double vb = isF ? 0.0 : (state < 0 ? lo : hi);
            double a[D::NV];
            double b = isF ? -svr : vb;
#pragma unroll
            for (int c = 0; c < D::NV; ++c) {
                const bool cF = (Fmask >> c) & 1ull;
                const double vbc = readlane_f64(vb, c);
                if (isF && !cF) b -= row[c] * vbc;
                a[c] = (isF && cF) ? row[c] : ((c == r && !isF) ? 1.0 : 0.0);
            }
            // Gaussian elimination without pivoting (SPD), pivot rows broadcast with v_readlane
            int bad = 0;
#pragma unroll
            for (int j = 0; j < D::NV; ++j) {
                const double piv = readlane_f64(a[j], j);
                bad |= !(piv > 0.0);
                const double f = (lane > j) ? a[j] / piv : 0.0;
                const double bj = readlane_f64(b, j);
                b -= f * bj;
#pragma unroll
                for (int c = j + 1; c < D::NV; ++c) {
                    const double pc = readlane_f64(a[c], j);
                    a[c] -= f * pc;
                }
            }
#pragma unroll
            for (int j = D::NV - 1; j >= 0; --j) {
                const double xj = readlane_f64(b, j) / readlane_f64(a[j], j);
                if (lane == j) v = xj;
                if (lane < j) b -= a[j] * xj;
            }
            if (bad) { status = VSMPC_STATUS_NUMERICAL; break; }
            double grad = svr;
#pragma unroll
            for (int c = 0; c < D::NV; ++c) grad += row[c] * readlane_f64(v, c);
            const double tolv = 1e-12 * (1.0 + fabs(v));
            const bool vlo = isF && (v < lo - tolv);
            const bool vhi = isF && (v > hi + tolv);
            const bool rlo = valid && state == -1 && !fixed && grad < -gtol;
            const bool rhi = valid && state == 1 && !fixed && grad > gtol;
            const bool inf = vlo || vhi || rlo || rhi;
            const unsigned long long imask = __ballot(inf);
            const int ninf = __popcll(imask);
            if (ninf == 0) { status = VSMPC_STATUS_SOLVED; break; }
            bool pick = inf;
            if (ninf < best) { best = ninf; patience = 3; }
            else if (patience > 0) { --patience; }
            else { pick = inf && (lane == 63 - __clzll(imask)); }  // least-index fallback (largest index)
            if (pick) state = vlo ? -1 : (vhi ? 1 : 0);
        }
        if (valid) {
            v = state < 0 ? lo : (state > 0 ? hi : v);  // bound variables sit exactly on their bound
            sV[lane] = v;
            sZ[D::NU + lane] = v;
        }
        if (lane == 0) { sFlags[1] = status; sFlags[2] = iters; }
    }
    __syncthreads();

    // ---------------------------------------------------------------- P5 back-substitution L^T z = y, v prescribed
    for (int p = D::NT - 1; p >= 0; --p) {
        if (wave == 0) {
            // lane j (mod 16) owns column j of L_pp
            const int j = lane & 15;
            const double* Tpp = sM + tile_off<D>(p, p);
            double colv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) colv[k] = (k >= j) ? Tpp[k * 17 + j] : 0.0;
            double w = sW[16 * p + j];
            double z = 0.0;
#pragma unroll
            for (int k = 15; k >= 0; --k) {
                const int gk = 16 * p + k;  // uniform
                double zk;
                if (gk >= D::NZ) zk = 0.0;
                else if (gk >= D::NU) zk = sZ[gk];
                else zk = readlane_f64(w, k) * sInvD[gk];
                if (j == k) z = zk;
                if (j < k) w -= colv[k] * zk;
            }
            if (lane < 16) sZ[16 * p + j] = z;
        }
        __syncthreads();
        if (p > 0) {
            if (tid < 16 * p) {
                const double* T = sM + tile_off<D>(p, tid >> 4) + (tid & 15);
                double acc2 = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) acc2 += T[k * 17] * sZ[16 * p + k];
                sW[tid] -= acc2;
            }
            __syncthreads();
        }
    }

    // ---------------------------------------------------------------- P6 forward simulation + outputs
    if (wave == 0) {
        const int r = lane < NX ? lane : NX - 1;
        double arow[NX], bjrow[NJ], btrow[NTH];
#pragma unroll
        for (int c = 0; c < NX; ++c) arow[c] = sA[r * NX + c];
#pragma unroll
        for (int c = 0; c < NJ; ++c) bjrow[c] = sBj[r * NJ + c];
#pragma unroll
        for (int c = 0; c < NTH; ++c) btrow[c] = sBt[r * NTH + c];
        const double cr = sC[r];
        double x = sIn[VSMPC_IN_X0 + r];
        if (lane < NX) sX[lane] = x;
        for (int k = 0; k < D::N; ++k) {
            const int jb = joint_block_of_stage<D>(k);
            const int tb = throttle_block_of_stage<D>(k);
            const int vq = tb == 0 ? D::NV - 4 : 4 * (tb - 1);  // internal offset of reference block tb
            double d = cr;
#pragma unroll
            for (int c = 0; c < NX; ++c) d += arow[c] * readlane_f64(x, c);
#pragma unroll
            for (int c = 0; c < NJ; ++c) d += bjrow[c] * sZ[NJ * jb + c];
#pragma unroll
            for (int c = 0; c < NTH; ++c) d += btrow[c] * sV[vq + c];
            x += cfg.dt[k] * d;
            if (lane < NX) sX[NX * (k + 1) + lane] = x;
        }
    }
    __syncthreads();

    if (xout != nullptr) {
        double* xo = xout + size_t(inst) * D::NVAR;
        for (int i = tid; i < D::NXS; i += BLOCK) xo[i] = sX[i];
        for (int i = tid; i < D::NU; i += BLOCK) xo[D::NXS + i] = sZ[i];
        if (tid < D::NV) {  // reference order v_0..v_{NVB-1}
            const int b = tid >> 2, c = tid & 3;
            const int q = b == 0 ? D::NV - 4 + c : 4 * (b - 1) + c;
            xo[D::NXS + D::NU + tid] = sV[q];
        }
    }
    if (fmout != nullptr && tid < VSMPC_FM_SIZE) {
        double v;
        if (tid < 8) v = sZ[tid];                                        // delta q           (variableSamplingMPC.cpp:99)
        else if (tid < 12) v = sV[D::NV - 4 + (tid - 8)];                // v0                (:100)
        else if (tid < 16) v = Jet::throttle_of_v(sV[D::NV - 4 + (tid - 12)]);  // throttle % (:146-149)
        else if (tid < 20) v = sX[NX + 12 + (tid - 16)];                 // thrust, node 1    (:101)
        else v = sX[NX + 16 + (tid - 20)];                               // thrust rate, node 1 (:102)
        fmout[size_t(inst) * VSMPC_FM_SIZE + tid] = v;
    }
    if (tid == 0) {
        int st = sFlags[1];
        if (sFlags[0]) st = VSMPC_STATUS_NUMERICAL;
        status_out[inst] = st;
        if (iters_out != nullptr) iters_out[inst] = sFlags[2];
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
template <class D>
static hipError_t launch_solve_t(const DevCfg& cfg, const double* d_in, int batch, double* d_x, double* d_fm,
                                 int* d_status, int* d_iters, double* dbgM, double* dbgL, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_kernel<D>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(Smem<D>::bytes));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(solve_kernel<D>, dim3(batch), dim3(BLOCK), Smem<D>::bytes, stream, cfg, d_in, batch, d_x,
                       d_fm, d_status, d_iters, dbgM, dbgL);
    return hipGetLastError();
}

template <class D>
static hipError_t launch_linearize_t(const DevCfg& cfg, const double* d_in, int batch, double* A, double* Bj,
                                     double* Bt, double* c, hipStream_t stream) {
    hipLaunchKernelGGL(linearize_kernel<D>, dim3(batch), dim3(128), 0, stream, cfg, d_in, A, Bj, Bt, c);
    return hipGetLastError();
}

using DimsPaper = Dims<17, 7, 12>;

int select_variant(int n_iter, int n_iter_small, int control_horizon) {
    if (n_iter == 17 && n_iter_small == 7 && control_horizon == 12) return VARIANT_PAPER;
    return VARIANT_NONE;
}

const char* variant_kernel_name(int variant) {
    switch (variant) {
        case VARIANT_PAPER: return "solve_kernel<Dims<17,7,12>>";
        default: return "none";
    }
}

int variant_condensed_dim(int variant) {
    switch (variant) {
        case VARIANT_PAPER: return DimsPaper::NP;
        default: return 0;
    }
}

hipError_t launch_solve(int variant, const DevCfg& cfg, const double* d_in, int batch, double* d_x, double* d_fm,
                        int* d_status, int* d_iters, double* dbgM, double* dbgL, hipStream_t stream) {
    switch (variant) {
        case VARIANT_PAPER:
            return launch_solve_t<DimsPaper>(cfg, d_in, batch, d_x, d_fm, d_status, d_iters, dbgM, dbgL, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_linearize(int variant, const DevCfg& cfg, const double* d_in, int batch, double* A, double* Bj,
                            double* Bt, double* c, hipStream_t stream) {
    switch (variant) {
        case VARIANT_PAPER: return launch_linearize_t<DimsPaper>(cfg, d_in, batch, A, Bj, Bt, c, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vsmpc
