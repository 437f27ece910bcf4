// Hand-written HIP kernels for gfx950 (MI355X): one workgroup per MPC instance.
//
// Replaces, per instance, IMPCProblem::update + IMPCProblem::solve + VariableSamplingMPC::solveMPC
// (IMPCProblem.cpp:150-298, variableSamplingMPC.cpp:88-112) with a structure-exploiting exact solve:
//
//   P0 linearise   A, Bj, Bt, c in LDS                       (systemDynamicsVSMPC.cpp:79-103,288-319,384-429)
//                  + joint reduction: a joint block acts on the dynamics through [Lambda_lin; Lambda_ang] (6 x 8) only, so
//                  SIX unknowns per block are condensed (Householder QR of (Lambda W^-1/2)^T, p0_joint_reduction); the two
//                  others have a closed form                  (constraintsVSMPC.cpp:85-103, costsVSMPC.cpp:375-381,564-591)
//   P1 condense    jet thrust sensitivities first (one two-state recursion per throttle column, four for the affine
//                  column).  Structured form (P1s, the default where Dims::STRUCT_P1): forward / adjoint recursions on
//                  three generator columns per joint block, the throttle columns and the affine column, one lane per
//                  (column, half), trajectory in registers; the tile entries of C formed on the matrix cores from the
//                  block sums the chains leave in LDS -- O(N^2) work.  SYRK form (every horizon; vsmpc_set_kernel_form):
//                  the sensitivity recursion in registers, thread (half, col) carries the linear-momentum half (p, h_lin,
//                  e_pos) or the angular half (rpy, h_ang, e_rpy) of one condensed column; two nodes (36 weighted rows =
//                  9 exact MFMA k-steps) per pass; C = sum_k Y_k^T Y_k with v_mfma_f64_16x16x4_f64, every pass a
//                  straight-line sequence of per-tile chains (compile-time slot count).  Accumulators in registers
//                  from here to P5; 256 threads, two workgroups per CU at the paper horizon
//                                                            (constraintsVSMPC.cpp:76-131, costsVSMPC.cpp:166-178)
//   P2 augment     M = C + R, gradient row                   (costsVSMPC.cpp:375-409,468-487,558-592)
//   P3 cholesky    right-looking LL^T on 16x16 tiles; the trailing matrix AND the finished factor stay in registers,
//                  LDS holds a ring of two panel columns + the throttle corner; trailing updates on MFMA.  A panel stream
//                  is generated assembly (vsmpc_panel_asm.inc <- tools/gen_panel_asm.py): every lane carries panel rows
//                  and, replicated per 16-lane row, a row of the diagonal tile; the pivot column is broadcast inside the FMA
//                  (v_fmac_f64 DPP row_newbcast).  Structured form: PIPELINED -- wavefront 0 factors panel p while
//                  wavefronts 1..3, which hold all tiles, apply panel p - 1 and invert diagonal tile p - 1 (cholesky_wave,
//                  TileTab<D, PIPE>); SYRK form: the panel shared by up to three wavefronts, all four update
//   P4 box QP      backward pass over the throttle tiles with only the hold pin; only if a bound is violated: block
//                  principal pivoting in one wavefront, dual form on P = X^T X for few violated bounds, primal form
//                  on the Schur complement otherwise, small systems in registers   (constraintsVSMPC.cpp:338-365)
//   P5 back-subst  joints from the register-resident factor, tile row by tile row: z_r = X_r^T (w_r - u_r), then every
//                  wavefront adds L_rq^T z_r of the tiles it owns to its partial sums u_q
//                  (pipelined schedule: the panel wavefront, which holds no tile, runs the jets link of P6 beside it)
//   P6 simulate    state trajectory, primal in the reference variable order, first-move block
//                                                            (variableSamplingMPC.cpp:93-108,138-151)
//
// FP64 throughout.  The un-condensed KKT system the reference hands to OSQP has condition number
// ~1e12; the condensed Hessian factored here is benign (1e2..1e3).
//
// Measured on MI355X (profiles/r01_microbench_*.txt, tools/microbench/lat_probe.hip): v_mfma_f64_16x16x4_f64 issues
// every 64 cycles per SIMD (77.7 TFLOP/s chip-wide, already with one wavefront per SIMD), a dependent one every ~95; FP64
// VALU work does not hide under it (shared FP64 datapath); a lone wavefront issues one FP64 vector instruction per ~5.7
// cycles whatever the dependencies (a serial stream costs its instruction count), a v_readlane takes ~32 cycles to land.
// Hence: every index is compile-time or scalar, LDS offsets are immediates, and the matrix-core streams carry nothing but
// operand loads.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "vsmpc_device.hpp"
#include "vsmpc_launch.hpp"

namespace vsmpc {

typedef double d4 __attribute__((ext_vector_type(4)));


template <class D, bool PIPE = false>
__device__ constexpr TileTab<D, PIPE> kTileTab{};
template <class D>
__device__ constexpr NactTab<D> kNactTab{};
template <class D>
__device__ constexpr TilePack<D> kTilePack{};

// compile-time loop: f(std::integral_constant<int, I>{}) for I = I0 .. N-1
template <int I, int N, class F>
VS_DEV void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// active accumulator slots of SYRK pass m, maximum over the wavefronts (see NactTab)
template <class D>
constexpr int nact_max(int m) {
    constexpr NactTab<D> t{};
    int n = 0;
    for (int w = 0; w < D::NWAVES; ++w) n = t.n[m][w] > n ? t.n[m][w] : n;
    return n;
}

// runs of consecutive SYRK passes with the same slot count and the same number of k-steps
template <class D>
struct PassGroups {
    static constexpr int NPASS = (D::N + 1) / 2;
    int start[NPASS], end[NPASS], nact[NPASS], nks[NPASS], n;
    constexpr PassGroups() : start{}, end{}, nact{}, nks{}, n(0) {
        for (int m = 0; m < NPASS; ++m) {
            const int a = nact_max<D>(m), k = (2 * m + 1 < D::N) ? 9 : 5;
            if (n > 0 && nact[n - 1] == a && nks[n - 1] == k) { end[n - 1] = m + 1; continue; }
            start[n] = m; end[n] = m + 1; nact[n] = a; nks[n] = k; ++n;
        }
    }
    static constexpr int count() { return PassGroups().n; }
};

VS_DEV double readlane_f64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// The lane id, re-derived where it is needed (two instructions) instead of carried there: a value that is live across a
// long phase is what the register allocator spills first, and at the 2x horizon every spilled dword is 4 MB of scratch
// traffic per 4096-instance launch.
VS_DEV int fresh_lane() {
    unsigned m = ~0u;
    asm volatile("" : "+s"(m));
    return int(__builtin_amdgcn_mbcnt_hi(m, __builtin_amdgcn_mbcnt_lo(m, 0u)));
}

// 1/sqrt(d) and 1/d: hardware seed + refinement (full double precision to ~2 ulp)
// The v_rsq_f64 / v_rcp_f64 seeds are good to 5e-8 (profiles/r01_microbench_rsq_accuracy.txt).  One cubically
// convergent (Halley) step y (1 + e/2 + 3 e^2/8), e = 1 - d y^2, reaches full precision in 5 instructions; two Newton
// steps need 7.  (With the pivots' reciprocal square roots on the critical path the Halley form measured slower; since
// the panel code overlaps them with the previous pivot's update, the instruction count is what matters.)
VS_DEV double fast_rsqrt(double d) {
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y, y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
}
VS_DEV double fast_rcp(double d) {
    double y = __builtin_amdgcn_rcp(d);
    y = y * fma(-d, y, 2.0);
    y = y * fma(-d, y, 2.0);
    return y;
}

// ------------------------------------------------------------------------------------------------
// LDS carve-up (doubles).  One region R is reused phase by phase:
//   P1      the Y buffer (two nodes x 18 weighted rows x YS)
//   P3      the ring of two panel columns + the throttle corner (Dims::L_TILES tiles, see vsmpc_device.hpp)
//   P4..P6  over the (by then dead) ring: box-QP work arrays, the per-wavefront partial sums of the register
//           back-substitution, the state trajectory and the stage forcing terms; the corner stays where it is
// Paper horizon: 71 KB in total, so two workgroups fit one CU.
// ------------------------------------------------------------------------------------------------
template <class D>
struct Smem {
    // the dual box QP and the chain-free first pass need the throttle block to span exactly two tile rows
    static constexpr bool DUALQP = D::NT - 2 == D::PVT && D::NU % 16 == 0 && D::NV >= 20 && D::NV <= 32;
    static constexpr int oIn = 0;
    static constexpr int oA = oIn + ((D::NIN + 3) & ~3);
    static constexpr int oBj = oA + NX * NX;
    static constexpr int oBt = oBj + NX * NJ;
    static constexpr int oC = oBt + NX * NTH;
    static constexpr int oVprev = oC + 28;
    static constexpr int oInvD = oVprev + 4;
    static constexpr int oW = oInvD + D::NP;
    static constexpr int oZ = oW + D::NP;
    static constexpr int oSvec = oZ + D::NP;
    static constexpr int oV = oSvec + D::NV;
    static constexpr int oDt = oV + D::NV;           // per-stage dt (copied out of the kernel arguments once)
    static constexpr int oCfg = oDt + MAX_STAGES;    // configuration scalars (CFG_* offsets)
    static constexpr int oFlags = oCfg + CFG_SIZE;   // 4 doubles worth of int flags
    // X_p = L_pp^-1 of the joint diagonal tiles and of the first throttle tile, produced by wavefronts that idle
    // during the panel factorisations of P3
    // joint reduction (p0_joint_reduction): the six Householder vectors, their betas, the reduced gradient Q^T b, the null
    // component n = -N^T b and W^(-1/2)
    static constexpr int oQR = (oFlags + 4 + 3) & ~3;
    static constexpr int QR_V = 0, QR_BETA = 48, QR_GY = 56, QR_NS = 62, QR_ISW = 64, QR_A = 72, QR_SIZE = 128;
    static constexpr int NXT = D::PVT + 1;
    static constexpr int oXinv = (oQR + QR_SIZE + 3) & ~3;
    static constexpr int oR = oXinv + NXT * D::TS;
    static constexpr int YROWS = 36;                 // two nodes x 18 weighted rows = 9 exact MFMA k-steps
    static constexpr int oY = oR;
    static constexpr int sizeY = YROWS * D::YS;
    static constexpr int oM = oR;                    // ring + corner tiles (Y is dead after P1)
    static constexpr int sizeM = D::L_TILES * D::TS;
    static constexpr int NVS = D::NV + 1;            // row stride of the box-QP work arrays
    static constexpr int oSv = oR;                   // Schur complement / columns of P
    // three throttle tile rows, tile aligned (the 2x horizon): dual form on a dense X assembled from tile products
    static constexpr bool DUAL3 = !DUALQP && D::NT - 3 == D::PVT && D::NU % 16 == 0 && D::NV > 32 && D::NV <= 48;
    static constexpr int oQP = oSv + D::NV * NVS;    // dual form: K | rows 16.. of X (DUALQP); K | X | two scratch tiles (DUAL3)
    static constexpr int sizeQP = DUALQP ? D::NV * NVS + (D::NV - 16) * NVS : (DUAL3 ? 2 * D::NV * NVS + 2 * D::TS : 0);
    static constexpr int oDual3T0 = oQP + 2 * D::NV * NVS;   // DUAL3: two tile-shaped scratches behind K and X
    static constexpr int oDual3T1 = oDual3T0 + D::TS;
    // per-wavefront partial sums of L^T z, NP each.  The box-QP arrays are dead by then: where the dense X would not fit
    // beside them (DUAL3) the two overlap
    static constexpr int oU = DUAL3 ? oR : oQP + sizeQP;
    static constexpr int oX = oU + D::NWAVES * D::NP;  // P6: state trajectory
    static constexpr int oF = oX + D::NXS;           // P6: per-stage input terms, NX per stage
    static constexpr int endScratch = (oF + NX * D::N > oQP + sizeQP) ? oF + NX * D::N : oQP + sizeQP;
    static_assert(endScratch <= oR + D::CORNER_TILE0 * D::TS, "P4..P6 scratch must not reach the corner tiles");
    // P1a: jet thrust trajectories [NJROW][N] and the affine column's momentum forcing [2][N][3], at the head of the X
    // region (the X tiles are not written before P3)
    static constexpr int NJROW = D::NV + NTH + 1;
    static constexpr int oJetT = oXinv;
    static constexpr int oGA = oJetT + NJROW * D::N;
    // P1s (structured condensing, Dims::STRUCT_P1): behind them, across the rest of the X region and R
    //   sH [2][NJPAIR][3][3]      block sums of H(i, i') over (row block, column block) pairs of the joint blocks
    //   sRb[2][NV + 1][HC][3]     W_c(i) of the throttle columns and of the affine column, summed over joint blocks
    //   sW3[2][NV + 1][N - 1][3]  W_c(i), i = 1 .. N - 1, of the throttle columns and of the affine column
    //   sAc[NV + 1][N - 1][4]     sum over the halves of A_mom[:, q]^T W_c(i) (formed by all wavefronts after the chains)
    //   sRefC[NREF][12]           reference window with the integrator offsets c_e folded into the x rows
    //   sZero                     zeros: what the columns without a thrust trajectory / forcing / reference read
    static constexpr int oSH = oGA + 6 * D::N;
    static constexpr int oSRb = oSH + 2 * D::NJPAIR * 9;
    static constexpr int oSW3 = oSRb + 2 * (D::NV + 1) * D::HC * 3;
    // long horizons (Dims::STRUCT_LONG) have no sW3: the chains add A_mom^T W straight into sAc (LDS atomics, two addends
    // per word: order independent).  A row of sAc keeps the stages i' = i - 1 >= ac_first(row) only: the tile columns at or
    // left of a throttle row before the v_0 block all start at stage NS or later, and tau_i = 0 up to a column's first stage
    static constexpr int oSAc = oSW3 + (D::STRUCT_LONG ? 0 : 2 * (D::NV + 1) * (D::N - 1) * 3);
    static constexpr int AC_SHORT = D::STRUCT_LONG ? D::NV - NTH : 0;   // rows [0, AC_SHORT) are short
    static constexpr int AC_NSH = D::STRUCT_LONG ? D::NS : 0;           // first stored i' of a short row
    VS_HD static constexpr int ac_first(int cr) { return cr < AC_SHORT ? AC_NSH : 0; }
    VS_HD static constexpr int ac_off(int cr) {   // offset of row cr's first stored stage (4 doubles per stage)
        // = cr < AC_SHORT ? cr (N - 1 - AC_NSH) 4 : (AC_SHORT (N - 1 - AC_NSH) + (cr - AC_SHORT) (N - 1)) 4, written without a
        // branch: as a conditional the compiler made basic blocks of it inside p1s_entries, and at their joins moved the
        // finished accumulator tiles through the vector registers (and a few values into scratch)
        return 4 * (cr * (D::N - 1) - (cr < AC_SHORT ? cr : AC_SHORT) * AC_NSH);
    }
    static constexpr int sizeAc = ac_off(D::NV + 1);
    static constexpr int oSRefC = oSAc + sizeAc;
    static constexpr int sizeZero = 3 * D::N > 12 * D::NREF ? 3 * D::N : 12 * D::NREF;
    // the zeros: long horizons borrow the (not yet used) w and z vectors of P3..P5
    static constexpr int oSZero = D::STRUCT_LONG ? oW : oSRefC + 12 * D::NREF;
    static_assert(!D::STRUCT_LONG || sizeZero <= 2 * D::NP, "zeros fit the w and z vectors");
    static constexpr int endP1s = D::STRUCT_LONG ? oSRefC + 12 * D::NREF : oSZero + sizeZero;
    // P3, pipelined schedule: one tile behind the ring and the corner -- the diagonal tile of the NEXT panel column as its holder
    // has it (updates of all earlier panels applied), parked there so that wavefront 0 can apply the current panel to it
    // itself the moment its stream ends (cholesky_wave).  The arrays of P1s that lie there are dead by then.
    static constexpr int oNextDiag = oM + sizeM;
    static constexpr int total_syrk = oR + (sizeY > sizeM ? sizeY : sizeM);
    static constexpr int total_struct = D::STRUCT_P1 ? (endP1s > oNextDiag + D::TS ? endP1s : oNextDiag + D::TS) : total_syrk;
    // both forms share one carve-up; a horizon with the structured form never launches the SYRK form unless asked to
    // (vsmpc_set_kernel_form), so each form gets its own size
    static constexpr int total = total_syrk;
    static constexpr size_t bytes = size_t(total) * sizeof(double);
    static constexpr size_t bytes_struct = size_t(total_struct) * sizeof(double);
    static_assert(bytes <= 160 * 1024 && bytes_struct <= 160 * 1024, "LDS budget of one CU");
    static_assert(D::WG_PER_CU < 2 || (bytes <= 80 * 1024 && bytes_struct <= 80 * 1024), "two workgroups per CU");
};

// tile (i, j), j <= i, of the factor in LDS: panel columns left of the throttle corner live in a ring of two
// (even columns at tile 0, odd ones at tile RING_A), the corner is dense behind the ring
template <class D>
VS_HD constexpr int tile_off_c(int i, int j) {
    return (j < D::PVT ? (j & 1) * D::RING_A + (i - j)
                       : D::CORNER_TILE0 + (i - D::PVT) * (i - D::PVT + 1) / 2 + (j - D::PVT)) * D::TS;
}
template <class D>
VS_DEV int tile_off(int i, int j) { return tile_off_c<D>(i, j); }

// element (gr, gc), gc <= gr, of a tile that is currently in LDS
template <class D>
VS_DEV int lower_at(int gr, int gc) {
    return tile_off<D>(gr >> 4, gc >> 4) + (gr & 15) * 17 + (gc & 15);
}

// ------------------------------------------------------------------------------------------------
// P0: linearisation into LDS (dense, row-major) — also the body of the linearise-only kernel
// ------------------------------------------------------------------------------------------------
// LAMBDA_BJ = false (the solve kernel): Bj is not filled with Lambda here -- p0_joint_reduction writes the reduced input
// matrix R^T into it instead.
template <class D, bool ZERO = true, bool SYNC = true, bool LAMBDA_BJ = true>
VS_DEV void p0_linearize(int use_jet, const double* __restrict__ sIn, double* __restrict__ sA,
                         double* __restrict__ sBj, double* __restrict__ sBt, double* __restrict__ sC,
                         double* __restrict__ sVprev, int tid, int nthreads) {
    if constexpr (ZERO) {
        for (int i = tid; i < NX * NX + NX * NJ + NX * NTH + 28; i += nthreads) sA[i] = 0.0;  // A,Bj,Bt,c contiguous
        __syncthreads();
    }
    // The independent pieces run in different wavefronts (0: attitude kinematics, 1: jets, 2: CoM / gravity, 2-3: copies)
    // so that their divergent paths overlap instead of serialising inside one wavefront; needs >= 256 threads.
    if (tid < 64) {
        // A[rpy, angMom] = W(rpy)^-1 * I_G^-1                       (systemDynamicsVSMPC.cpp:86-87,140-147)
        // One wavefront: even lanes take sin / cos of the roll, odd lanes of the pitch (ONE sincos instead of two in a row:
        // it is the longest dependent chain of P0), lanes 0..8 then form one entry (i, j) each.
        const double* I = sIn + VSMPC_IN_INERTIA;
        const double a = I[0], b = I[1], c = I[2], d = I[3], e = I[4], f = I[5], g = I[6], h = I[7], k = I[8];
        double sn, cs;
        sincos(sIn[VSMPC_IN_RPY + (tid & 1)], &sn, &cs);
        const double A00 = e * k - f * h, A01 = c * h - b * k, A02 = b * f - c * e;
        const double A10 = f * g - d * k, A11 = a * k - c * g, A12 = c * d - a * f;
        const double A20 = d * h - e * g, A21 = b * g - a * h, A22 = a * e - b * d;
        const double idet = fast_rcp(a * A00 + b * A10 + c * A20);
        const double sr = readlane_f64(sn, 0), cr = readlane_f64(cs, 0), sp = readlane_f64(sn, 1), cp = readlane_f64(cs, 1);
        const double icp = fast_rcp(cp), tp = sp * icp;
        const int i = tid / 3, j = tid - 3 * i;                      // entry (i, j), tid < 9
        // row i of W^-1 = [1, sr tp, cr tp; 0, cr, -sr; 0, sr / cp, cr / cp], column j of I^-1 = adj[:, j] / det
        const double w0 = i == 0 ? 1.0 : 0.0;
        const double w1 = i == 0 ? sr * tp : (i == 1 ? cr : sr * icp);
        const double w2 = i == 0 ? cr * tp : (i == 1 ? -sr : cr * icp);
        const double c0 = j == 0 ? A00 : (j == 1 ? A01 : A02);
        const double c1 = j == 0 ? A10 : (j == 1 ? A11 : A12);
        const double c2 = j == 0 ? A20 : (j == 1 ? A21 : A22);
        if (tid < 9) sA[(6 + i) * NX + 9 + j] = (w0 * c0 + w1 * c1 + w2 * c2) * idet;
    } else if (tid >= 64 && tid < 68) {
        // jets                                                      (systemDynamicsVSMPC.cpp:384-429)
        const int i = tid - 64;
        sVprev[i] = Jet::v_of_throttle_div(sIn[VSMPC_IN_UPREV + i]);
        if (use_jet) {
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
    } else if (tid == 128) {
        // CoM kinematics, -S(omega) blocks, gravity term, integrators  (systemDynamicsVSMPC.cpp:90-91,296-316)
        const double m = sIn[VSMPC_IN_MASS], im = fast_rcp(m);
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
    } else if (tid >= 192 && tid < 216) {
        // thrust maps A[linMom|angMom, T] = A_mom,body                (systemDynamicsVSMPC.cpp:92-93,303-304)
        const int e = tid - 192, r = e >> 2, j = e & 3;  // r in 0..5
        const int row = r < 3 ? 3 + r : 6 + r;         // 3..5, 9..11
        sA[row * NX + 12 + j] = sIn[VSMPC_IN_AMOM + e];
    } else if (LAMBDA_BJ && tid >= 216 && tid < 240) {
        const int e = tid - 216, r = e >> 3, j = e & 7;  // Lambda_lin,B -> Bj[3..5]   (:305-306)
        sBj[(3 + r) * NJ + j] = sIn[VSMPC_IN_LLIN + e];
    } else if (LAMBDA_BJ && tid >= 136 && tid < 160) {
        const int e = tid - 136, r = e >> 3, j = e & 7;  // Lambda_ang,B -> Bj[9..11]  (:94-95)
        sBj[(9 + r) * NJ + j] = sIn[VSMPC_IN_LANG + e];
    }
    if constexpr (SYNC) __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Joint reduction (see NJC in vsmpc_device.hpp), by ONE wavefront.  Householder QR of A = (Lambda W^(-1/2))^T (8 x 6):
// lane c < 6 carries column c of A (= row c of [Lambda_lin; Lambda_ang], scaled), lane 6 the vector b = w_reg W^(-1/2)
// q_err that rides along; step k takes its reflector from lane k (v_readlane broadcasts) and every lane behind applies
// it to its own column.  Leaves in LDS: R^T as the joint input matrix (rows 3..5 and 9..11 of Bj, columns 0..5), the
// reflectors v_k and beta_k (for the way back in P6), Q^T b (the gradient of the reduced unknowns), n = -N^T b and
// W^(-1/2).  No inverse, no division by a pivot: a rank-deficient Lambda (no thrust) leaves zero columns in R^T.
// ------------------------------------------------------------------------------------------------
// Steps K0 .. K1 - 1 of the six; a wavefront that does not start at 0 picks the columns up from LDS where the previous
// one left them (QR_A), so that the work can be spread over the idle stretches of different wavefronts: the first half in
// wavefront 3 during P0 (which has only copies to do there), the second half in a generator wavefront while it waits for
// the throttle chains -- a lone wavefront needs ~900 cycles per step (two reductions, a reciprocal square root and a
// reciprocal in one dependent chain).
template <class D, int K0 = 0, int K1 = NJC>
VS_DEV void p0_joint_reduction(double* __restrict__ sm, int lane) {
    using S = Smem<D>;
    static_assert(VSMPC_IN_LANG == VSMPC_IN_LLIN + 24, "Lambda_lin and Lambda_ang are adjacent in the record");
    const double* sIn = sm + S::oIn;
    const double* sCfg = sm + S::oCfg;
    double* sBj = sm + S::oBj;
    double* sQR = sm + S::oQR;
    const int c = lane < 6 ? lane : 6;   // lanes beyond 6 shadow lane 6 and store nothing
    double a[8];
    if constexpr (K0 == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double isw = fast_rsqrt(sCfg[CFG_WJ + j]);                       // uniform
            const double lam = sIn[VSMPC_IN_LLIN + (c < 6 ? c : 0) * 8 + j];
            const double bq = sCfg[CFG_WREG] * sIn[VSMPC_IN_QERR + j];             // costsVSMPC.cpp:586-590
            a[j] = (c < 6 ? lam : bq) * isw;
            if (lane == 0) sQR[S::QR_ISW + j] = isw;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = sQR[S::QR_A + 8 * c + j];
    }
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        // (sums as two or three partial chains: the dependent length is what a lone wavefront pays for)
        double s0 = a[k] * a[k], s1 = 0.0;
#pragma unroll
        for (int i = k + 1; i < 8; i += 2) {
            s1 = fma(a[i], a[i], s1);
            if (i + 1 < 8) s0 = fma(a[i + 1], a[i + 1], s0);
        }
        const double s = s0 + s1;
        const double sk = readlane_f64(s, k), xk = readlane_f64(a[k], k);      // the pivot column's, wave uniform
        const bool nz = sk > 1e-300;
        const double rs = fast_rsqrt(nz ? sk : 1.0);
        const double nrm = nz ? sk * rs : 0.0;
        const double alpha = xk >= 0.0 ? -nrm : nrm;
        const double beta = nz ? rs * fast_rcp(nrm + fabs(xk)) : 0.0;          // 1 / (nrm (nrm + |x_k|))
        double v[8];
        v[k] = xk - alpha;
#pragma unroll
        for (int i = k + 1; i < 8; ++i) v[i] = readlane_f64(a[i], k);
        double w0 = v[k] * a[k], w1 = 0.0;
#pragma unroll
        for (int i = k + 1; i < 8; i += 2) {
            w1 = fma(v[i], a[i], w1);
            if (i + 1 < 8) w0 = fma(v[i + 1], a[i + 1], w0);
        }
        const double w = (w0 + w1) * beta;
        const bool behind = lane > k;
#pragma unroll
        for (int i = k; i < 8; ++i) a[i] = behind ? fma(-w, v[i], a[i]) : a[i];
        a[k] = lane == k ? alpha : a[k];                                        // R[k][k]
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) sQR[S::QR_V + 8 * k + i] = i >= k ? v[i] : 0.0;
            sQR[S::QR_BETA + k] = beta;
        }
    }
    if constexpr (K1 < NJC) {
        if (lane < 7) {
#pragma unroll
            for (int j = 0; j < 8; ++j) sQR[S::QR_A + 8 * lane + j] = a[j];
        }
        return;
    }
    if (lane < 6) {   // column `lane` of R = row `lane` of the input matrix R^T: entries k <= lane
        const int row = lane < 3 ? 3 + lane : 6 + lane;                         // momentum rows 3..5, 9..11
#pragma unroll
        for (int k = 0; k < NJC; ++k) sBj[row * NJ + k] = k <= lane ? a[k] : 0.0;
    } else if (lane == 6) {
#pragma unroll
        for (int k = 0; k < NJC; ++k) sQR[S::QR_GY + k] = a[k];
        sQR[S::QR_NS + 0] = -a[6];
        sQR[S::QR_NS + 1] = -a[7];
    }
}

// U = W^(-1/2) H_1 ... H_6 [y; n] for one joint block (the way back from the reduced unknowns).  Resumable: reflectors
// K1 - 1 down to K0; K1 = NJC loads [y; n], K0 = 0 ends with the scaling (P6 runs the first half beside its input-term pass)
template <class D, int K1 = NJC, int K0 = 0>
VS_DEV void joint_expand(const double* __restrict__ sQR, const double* __restrict__ y, double (&u)[8]) {
    using S = Smem<D>;
    if constexpr (K1 == NJC) {
#pragma unroll
        for (int i = 0; i < NJC; ++i) u[i] = y[i];
        u[6] = sQR[S::QR_NS + 0];
        u[7] = sQR[S::QR_NS + 1];
    }
#pragma unroll
    for (int k = K1 - 1; k >= K0; --k) {
        double t = 0.0;
#pragma unroll
        for (int i = k; i < 8; ++i) t = fma(sQR[S::QR_V + 8 * k + i], u[i], t);   // uniform addresses: LDS broadcasts
        t *= sQR[S::QR_BETA + k];
#pragma unroll
        for (int i = k; i < 8; ++i) u[i] = fma(-t, sQR[S::QR_V + 8 * k + i], u[i]);
    }
    if constexpr (K0 == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] *= sQR[S::QR_ISW + i];
    }
}

// ------------------------------------------------------------------------------------------------
// linearise-only kernel (vsmpc_linearize_batch)
// ------------------------------------------------------------------------------------------------
template <class D>
__global__ __launch_bounds__(256) void linearize_kernel(DevCfg cfg, const double* __restrict__ in,
                                                         double* __restrict__ A, double* __restrict__ Bj,
                                                         double* __restrict__ Bt, double* __restrict__ c) {
    __shared__ double sIn[(D::NIN + 3) & ~3];
    __shared__ double sLin[NX * NX + NX * NJ + NX * NTH + 28 + 4];
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int i = tid; i < D::NIN; i += 256) sIn[i] = in[size_t(b) * D::NIN + i];
    __syncthreads();
    double* sA = sLin;
    double* sBj = sA + NX * NX;
    double* sBt = sBj + NX * NJ;
    double* sC = sBt + NX * NTH;
    double* sVprev = sC + 28;
    p0_linearize<D>(cfg.use_jet, sIn, sA, sBj, sBt, sC, sVprev, tid, 256);
    for (int i = tid; i < NX * NX; i += 256) A[size_t(b) * NX * NX + i] = sA[i];
    for (int i = tid; i < NX * NJ; i += 256) Bj[size_t(b) * NX * NJ + i] = sBj[i];
    for (int i = tid; i < NX * NTH; i += 256) Bt[size_t(b) * NX * NTH + i] = sBt[i];
    for (int i = tid; i < NX; i += 256) c[size_t(b) * NX + i] = sC[i];
}

// ------------------------------------------------------------------------------------------------
// cost terms on the condensed inputs (P2)
// ------------------------------------------------------------------------------------------------
// (sGy = Q^T b of the joint reduction; the reduced joint unknowns and the dummies have unit weights)
template <class D>
VS_DEV double input_cost_term(const double* __restrict__ sCfg, const double* __restrict__ sGy,
                              const double* __restrict__ sVprev, int gr, int gc) {
    if (gr < D::NU) return (gr == gc) ? 1.0 : 0.0;   // |y|^2 / 2 = U^T W U / 2  (costsVSMPC.cpp:375-381,564-571)
    if (gr < D::NZ) {
        if (gc < D::NU) return 0.0;
        const int q1 = gr - D::NU, q2 = gc - D::NU;
        if ((q1 & 3) != (q2 & 3)) return 0.0;
        const int b1 = v_block_of_internal<D>(q1), b2 = v_block_of_internal<D>(q2);
        if (b1 == b2)  // first-difference penalty + v0 anchor  (costsVSMPC.cpp:383-409,472-476)
            return sCfg[CFG_WTHR] * double((b1 > 0) + (b1 < D::NVB - 1)) + (b1 == 0 ? sCfg[CFG_WINIT] : 0.0);
        const int db = b1 - b2;
        return (db == 1 || db == -1) ? -sCfg[CFG_WTHR] : 0.0;
    }
    if (gr == D::NZ && gc < D::NZ) {  // gradient row
        if (gc < D::NU) return gc < D::NUY ? sGy[gc % NJC] : 0.0;                   // costsVSMPC.cpp:586-590, reduced
        const int q = gc - D::NU;
        return v_block_of_internal<D>(q) == 0 ? -sCfg[CFG_WINIT] * sVprev[q & 3] : 0.0;  // costsVSMPC.cpp:479-485
    }
    return 0.0;
}

// ------------------------------------------------------------------------------------------------
// SYRK of P1, one accumulator tile (slot) at a time: NKS k-steps of 4 rows as ONE dependent chain on the tile's
// accumulator (a dependent v_mfma_f64_16x16x4_f64 issues every 64 cycles, like independent ones).
//   * The slots a pass runs are a prefix NACT-1, ..., 0 of the stage-sorted tile table, and NACT is a COMPILE-TIME
//     constant of the pass, the same for the four wavefronts (the maximum over them; a wavefront with fewer active
//     tiles multiplies columns of Y that are still exactly zero).  The chains of a pass are therefore straight-line
//     code.  Every earlier form with control flow around the chains (an instantiation per slot count through v9, a
//     fall-through switch, a branch per slot) made the register allocator move whole accumulator tiles at the joins
//     and the loop back-edge: ~1.1k cycles per pass whatever the number of chains (38.8k cycles of matrix-core
//     section against a floor of 29.4k; now 31.6k).  Short horizons unroll the pass loop, long ones run one rolled
//     loop per distinct slot count (PassGroups).
//   * Operand loads are software-pipelined SYRK_DIST instructions ahead ACROSS slots and pinned with sched_barrier:
//     the wave's stream blocks at every MFMA issue until the pipe is free (64 cycles), an LDS read returns in ~130.
//     `ha`/`hb` carry the first SYRK_DIST operand pairs of the slot in and those of the NEXT slot (slot q - 1) out.
// Build-time switches (measurement variants, tools/exp_build.sh): VS_SYRK_DIST prefetch distance, VS_SYRK_TIED inline
// assembly with a tied accumulator, VS_SYRK_UNROLL / VS_UNROLL_TPW unrolled passes up to that many slots per wavefront,
// VS_DUAL3_MAX / VS_KMID the box QP's dual-form threshold and register-solver size at long horizons.
// ------------------------------------------------------------------------------------------------
#ifndef VS_SYRK_DIST
#define VS_SYRK_DIST 2
#endif
#ifndef VS_SYRK_TIED
#define VS_SYRK_TIED 0
#endif
#ifndef VS_UNROLL_TPW
#define VS_UNROLL_TPW 12
#endif
#ifndef VS_SYRK_UNROLL
#define VS_SYRK_UNROLL 1
#endif
constexpr int SYRK_DIST = VS_SYRK_DIST;
#ifndef VS_KMID
#define VS_KMID 24
#endif


// TIED: the matrix instruction is written as inline assembly whose accumulator is a read-write operand, i.e. the result
// lands in the registers the tile already occupies.  With the builtin the register allocator is free to put the result
// of a chain's first instruction into fresh registers, and at the control-flow joins around the chains it then moves
// whole tiles back (measured: ~1.1k cycles per pass, whatever the number of chains).  The compiler does not know that the
// statement is a matrix instruction, so the software wait states between it and a vector instruction that touches the
// tile are placed by hand: 18 behind the last instruction of a chain (CDNA3 ISA 4.5: DGEMM 16x16x4 result -> VALU
// read/write), which cost nothing -- the matrix pipe is busy for 64 cycles with that instruction anyway -- and 2 in
// front of the first one (vector write -> matrix read).  Within a chain the accumulator forwards back to back.
template <class D, int NKS, bool TIED = false, bool PIN = true>
VS_DEV void syrk_slot(d4& acc, const double* __restrict__ pa, const double* __restrict__ pb, double (&ha)[SYRK_DIST],
                      double (&hb)[SYRK_DIST], const double* __restrict__ pan, const double* __restrict__ pbn) {
    static_assert(NKS >= 2 * SYRK_DIST - 1, "pipeline depth");
    double av[NKS], bv[NKS], na[SYRK_DIST], nb[SYRK_DIST];
#pragma unroll
    for (int ks = 0; ks < SYRK_DIST; ++ks) { av[ks] = ha[ks]; bv[ks] = hb[ks]; }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        if constexpr (TIED) {
            if (ks == 0)
                asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(av[ks]), "v"(bv[ks]));
            else if (ks == NKS - 1)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 1"
                             : "+v"(acc) : "v"(av[ks]), "v"(bv[ks]));
            else
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(av[ks]), "v"(bv[ks]));
        } else {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], bv[ks], acc, 0, 0, 0);
        }
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
        const int n = ks + SYRK_DIST;
        if (n < NKS) {
            av[n] = pa[n * 4 * D::YS];
            bv[n] = pb[n * 4 * D::YS];
        }
        // the head of the NEXT slot is requested behind the FIRST instructions of this chain, not the last ones: by the
        // control-flow join that follows the chain every load has long returned (the compiler drains the LDS counter at
        // a join: it cannot count outstanding loads across predecessors)
        if (ks < SYRK_DIST) {
            na[ks] = pan[ks * 4 * D::YS];
            nb[ks] = pbn[ks * 4 * D::YS];
        }
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < SYRK_DIST; ++ks) { ha[ks] = na[ks]; hb[ks] = nb[ks]; }
}

// ------------------------------------------------------------------------------------------------
// P3 panel factorisation by ONE wavefront, branch-free: lane l owns panel rows 16p + 64 s + l for the row slots
// s = 0..NSLOT-1 (slot 0, lanes 0..15 = the diagonal tile).  Pivot-column entries are broadcast with v_readlane;
// each broadcast feeds the updates of all slots.  NPIV < 16 only for the last panel, whose remaining rows
// (gradient row, padding) are carried along as ordinary panel rows.  `Lb` is the LDS tile storage of the panel
// column (ring of two columns + throttle corner, tile_off_c).
// Returns non-zero if a pivot was not positive.
// ------------------------------------------------------------------------------------------------
// SPLIT: the panel is shared by several wavefronts with no communication.  Each takes the diagonal tile in lanes
// 0..15 of its first slot (factored redundantly, bit-identical everywhere) and 64 NSLOT - 16 of the rows below it in
// the remaining lanes (48 rows and the short one-slot stream wherever three wavefronts cover the panel).  The other wavefronts read the unfactored diagonal tile while wavefront
// w = 0 works, so in SPLIT mode the factored diagonal tile is not stored here: it is handed back in `diag` (lanes
// 0..15 of wavefront 0) and stored by the caller after the workgroup barrier that ends the panel step.
template <class D, int NSLOT, int NPIV, bool SPLIT = false, bool PLDS = false>
VS_DEV int panel_factor(double* __restrict__ Lb, double* __restrict__ sInvD, int p, int lane, int w, double (&diag)[16],
                         double* sCol) {
    constexpr int RPW = 64 * NSLOT - 16;  // SPLIT: rows below the diagonal tile carried by one wavefront
    double* T[NSLOT];
    bool ok[NSLOT];
    double a[NSLOT][16];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int r = SPLIT ? ((s == 0 && lane < 16) ? 16 * p + lane : 16 * p + RPW * w + 64 * s + lane)
                            : 16 * p + 64 * s + lane;
        ok[s] = r < D::NP;
        T[s] = Lb + tile_off<D>(ok[s] ? (r >> 4) : p, p) + (r & 15) * 17;  // rows beyond the matrix read an
#pragma unroll                                                              // in-range tile and are never stored
        for (int c = 0; c < 16; ++c) a[s][c] = T[s][c];
    }
    double dmin = 1.0;     // all pivots positive <=> min(pivots) > 0; a NaN pivot (fmin skips it) makes every later
                           // pivot and the last reciprocal square root NaN, which is checked at the end
    double inv_mine = 1.0, inv_last = 1.0;   // lane j keeps 1/L_jj (a select per pivot, no branch on the pivot chain)
    // software-pipelined pivots: the next pivot is complete as soon as the first column of this pivot's update is
    // done, so its reciprocal square root (a ~75-cycle dependent chain) is issued there and overlaps the rest of the
    // update instead of following it
    double d = readlane_f64(a[0][0], 0);
    double inv = fast_rsqrt(d);
#pragma unroll
    for (int j = 0; j < NPIV; ++j) {
        dmin = fmin(dmin, d);
        inv_mine = lane == j ? inv : inv_mine;
        inv_last = inv;
        double l[NSLOT];
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) { l[s] = a[s][j] * inv; a[s][j] = l[s]; }
        if (j + 1 < 16) {
            const double lcj = readlane_f64(l[0], j + 1);
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) a[s][j + 1] = fma(-l[s], lcj, a[s][j + 1]);
            if (j + 1 < NPIV) {
                d = readlane_f64(a[0][j + 1], j + 1);
                inv = fast_rsqrt(d);
            }
        }
        // PLDS (the form for two workgroups per CU): the pivot column reaches the columns c >= j + 2 through LDS -- every lane
        // stores its slot-0 entry (entries 0..15 of the wavefront's 64-double strip are the rows of the diagonal tile), the
        // updates read entry c with a wave-uniform address: one (often half an) instruction per column instead of a
        // v_readlane pair.  Fewer instructions, more latency: with a second workgroup on the CU to fill the waits it is
        // 2.7 % faster (batch 4096: 409.6 -> 398.6 us), a lone workgroup is 0.5 % slower and the 2x horizon 9 % slower, so
        // the launcher picks it by batch size.  Only column j + 1, which the next pivot waits for, always goes through v_readlane.
        if (PLDS && j + 2 < 16) sCol[lane] = l[0];
#pragma unroll
        for (int c = j + 2; c < 16; ++c) {
            const double lcj = PLDS ? sCol[c] : readlane_f64(l[0], c);
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) a[s][c] = fma(-l[s], lcj, a[s][c]);
            // where a lane carries two rows the column's updates pass through an (empty) volatile statement: volatile
            // statements keep their order, so both row slots are updated while the broadcast is in its scalar registers.
            // Otherwise the second slot's updates are postponed behind later pivots and every broadcast waits for them in a
            // vector lane (v_writelane / v_readlane pairs: +30 % instructions in the panel stream of the 2x horizon)
            if constexpr (NSLOT == 2) asm volatile("" : "+v"(a[0][c]), "+v"(a[1][c]));
        }
    }
    if (ok[0] && (!SPLIT || lane >= 16)) {  // above its diagonal the diagonal tile holds leftovers: readers mask it
#pragma unroll
        for (int c = 0; c < 16; ++c) T[0][c] = a[0][c];
    }
    if constexpr (SPLIT) {
#pragma unroll
        for (int c = 0; c < 16; ++c) diag[c] = a[0][c];
    }
#pragma unroll
    for (int s = 1; s < NSLOT; ++s)
        if (ok[s]) {
#pragma unroll
            for (int c = 0; c < 16; ++c) T[s][c] = a[s][c];
        }
    if (lane < NPIV && (!SPLIT || w == 0)) sInvD[16 * p + lane] = inv_mine;
    return !(dmin > 0.0) || !(inv_last == inv_last);
}

// ------------------------------------------------------------------------------------------------
// The panel streams as hand-scheduled assembly with DPP broadcasts (kernel v26; tools/gen_panel_asm.py has the why: a lone
// wavefront issues one FP64 instruction per ~5.5 cycles whatever the dependencies, so a stream costs its instruction
// count, and v_fmac_f64 with DPP row_newbcast needs two instructions per updated column where v_readlane needs three).
// Lane 16 r + c of wavefront w carries panel row 16 p + 16 + 64 w + 16 r + c -- all 64 lanes carry rows BELOW the diagonal
// tile -- and, in a second set of registers, row c of the diagonal tile, which every 16-lane row factors redundantly
// (bit-identical in all rows and wavefronts).  Same arithmetic as panel_factor, operation by operation; in isolation
// 2.7 k cycles against 3.9 k (tools/microbench/panel_probe.hip, profiles/r04_microbench_panel_probe.txt).
// VS_PANEL_DPP=0 builds the C++ streams instead (A/B, and the reference the assembly is tested against).
// ------------------------------------------------------------------------------------------------
#ifndef VS_PANEL_DPP
#define VS_PANEL_DPP 1
#endif
#include "vsmpc_panel_asm.inc"
VS_DEV unsigned lds_addr(const double* q) { return unsigned(reinterpret_cast<uintptr_t>(q)); }   // flat -> LDS byte address

// Panel p < NT - 1, wavefront w of those that share it, S row slots per lane: rows 16 p + 16 + 64 (S w + s) + lane of the
// panel column; the factored diagonal tile comes back in `diag` (lanes 0..15) and 1 / L_jj goes to sInvD from wavefront 0.
// `scratch` = 32 doubles of this wavefront nobody reads: rows beyond the matrix and the other wavefronts' 1 / L_jj end there.
// Returns non-zero if a pivot was not positive (its reciprocal square root is NaN, and then so is everything after it down
// to the last one).
// KB > 0: the variant with a workgroup barrier inside (s_barrier behind pivot KB of the diagonal tile, the rows below are
// loaded behind it): the caller's other wavefronts execute a matching __syncthreads().
template <class D, int S = 1, int KB = 0>
VS_DEV int panel_dpp(double* __restrict__ Lb, double* __restrict__ sInvD, int p, int lane, int w, double (&diag)[16],
                      double* scratch) {
    unsigned ld[S], st[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int r = 16 * p + 16 + 64 * (S * w + s) + lane;
        const bool ok = r < D::NP;
        ld[s] = lds_addr(Lb + tile_off<D>(ok ? (r >> 4) : p, p) + (r & 15) * 17);
        st[s] = ok ? ld[s] : lds_addr(scratch);
    }
    const unsigned dg = lds_addr(Lb + tile_off<D>(p, p) + (lane & 15) * 17);
    const unsigned iv = lds_addr(w == 0 ? sInvD + 16 * p : scratch + 16);
    double inv_last;
    static_assert(S >= 1 && S <= 3, "tools/gen_panel_asm.py generates one, two and three row slots");
    static_assert(KB == 0 || (S == 1 && KB == 3) || (S == 2 && (KB == 3 || KB == 6)) || (S == 3 && KB == 6), "tools/gen_panel_asm.py BARRIER_VARIANTS");
    if constexpr (S == 1 && KB == 0) panel16x1_dpp(ld[0], st[0], dg, iv, diag, inv_last);
    else if constexpr (S == 2 && KB == 0) panel16x2_dpp(ld[0], st[0], ld[1], st[1], dg, iv, diag, inv_last);
    else if constexpr (S == 3 && KB == 0) panel16x3_dpp(ld[0], st[0], ld[1], st[1], ld[2], st[2], dg, iv, diag, inv_last);
    else if constexpr (S == 1) panel16x1_b3_dpp(ld[0], st[0], dg, iv, diag, inv_last);
    else if constexpr (S == 2 && KB == 3) panel16x2_b3_dpp(ld[0], st[0], ld[1], st[1], dg, iv, diag, inv_last);
    else if constexpr (S == 2) panel16x2_b6_dpp(ld[0], st[0], ld[1], st[1], dg, iv, diag, inv_last);
    else panel16x3_b6_dpp(ld[0], st[0], ld[1], st[1], ld[2], st[2], dg, iv, diag, inv_last);
    return !(inv_last == inv_last);
}
// The last panel (one wavefront): NPIV pivots, the remaining rows of the tile (gradient row, padding) are ordinary rows.
template <class D, int NPIV>
VS_DEV int panel_last_dpp(double* __restrict__ Lb, double* __restrict__ sInvD, int lane) {
    constexpr int p = D::NT - 1;
    double g[16], inv_last;
    double* T = Lb + tile_off<D>(p, p) + (lane & 15) * 17;
    static_assert(NPIV == 8 || NPIV == 12, "tools/gen_panel_asm.py LAST_PANEL_PIVOTS");
    if constexpr (NPIV == 8) panel_last8_dpp(lds_addr(T), lds_addr(sInvD + 16 * p), g, inv_last);
    else panel_last12_dpp(lds_addr(T), lds_addr(sInvD + 16 * p), g, inv_last);
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) T[c] = g[c];
    }
    return !(inv_last == inv_last);
}

// Long horizons (one wavefront per SIMD, 512 registers per lane): the accumulator tiles belong in the AGPR half of the
// register file for all of P2..P5.  Left to itself the allocator parked about half of the 30 tiles of a wavefront in scratch
// and reloaded them around every trailing update (1.3 GB of scratch writes per 4096-instance launch of the 2x horizon,
// profiles/r03_v14_c4_h2x4096_summary.md).  An empty asm statement with "a" constraints on fifteen tiles at once (an asm
// statement takes thirty operands, a read-write one counts twice) says so at the phase boundaries.
template <int TPW>
VS_DEV void pin_tiles_agpr(d4 (&acc)[TPW]) {
    constexpr int G = 15;
#pragma unroll
    for (int b = 0; b + G <= TPW; b += G)
        asm volatile("" : "+a"(acc[b]), "+a"(acc[b + 1]), "+a"(acc[b + 2]), "+a"(acc[b + 3]), "+a"(acc[b + 4]), "+a"(acc[b + 5]),
                          "+a"(acc[b + 6]), "+a"(acc[b + 7]), "+a"(acc[b + 8]), "+a"(acc[b + 9]), "+a"(acc[b + 10]),
                          "+a"(acc[b + 11]), "+a"(acc[b + 12]), "+a"(acc[b + 13]), "+a"(acc[b + 14]));
#pragma unroll
    for (int q = (TPW / G) * G; q < TPW; ++q) asm volatile("" : "+a"(acc[q]));
}

// ------------------------------------------------------------------------------------------------
// P2 + P3 for wavefront W, straight-line: the panel index and the tile table are compile-time, so every
// "does this tile take part" decision folds away and every LDS offset is an immediate.
//   P2  input-cost terms (joint weights, throttle coupling, gradient row) are added to the SYRK
//       accumulators in registers;
//   P3  right-looking blocked Cholesky with a register-resident trailing matrix AND a register-resident factor:
//       a tile goes to LDS exactly once, when its tile column becomes the panel (ring of two columns, see
//       Dims); the panel is factored in LDS (panel_factor); every wavefront updates the tiles it owns with four
//       v_mfma_f64_16x16x4_f64 per tile and takes the finished tiles of the panel column it owns BACK into the
//       accumulator registers that held them, where the back-substitution of P5 finds them.
// All instantiations execute the same number of workgroup barriers.
// ------------------------------------------------------------------------------------------------
// X = L_pp^-1 of one factored diagonal tile by one wavefront: lane j carries column j (lanes >= 16 shadow), the
// entries of L_pp and 1/L_ii are wave-uniform LDS broadcasts.  X is stored like a tile: X[i][j] at i*17 + j.
template <class D>
VS_DEV void tile_inverse(const double* __restrict__ Lpp, const double* __restrict__ invd, double* __restrict__ X,
                         int lane) {
    const int j = lane & 15;
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k + 1 < i; k += 2) {
            s0 = fma(Lpp[i * 17 + k], x[k], s0);
            s1 = fma(Lpp[i * 17 + k + 1], x[k + 1], s1);
        }
        if (i & 1) s0 = fma(Lpp[i * 17 + i - 1], x[i - 1], s0);
        const double di = invd[i];
        x[i] = (i == j) ? di : -di * (s0 + s1);  // rows above the diagonal come out as (signed) zeros
    }
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) X[i * 17 + j] = x[i];
    }
}

// rows I0 .. I1 - 1 of tile_inverse, resumable: x carries the column of this lane between calls (rows < I0 done before)
template <int I0, int I1>
VS_DEV void tile_inverse_rows(const double* __restrict__ Lpp, const double* __restrict__ invd, double (&x)[16], int lane) {
    const int j = lane & 15;
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k + 1 < i; k += 2) {
            s0 = fma(Lpp[i * 17 + k], x[k], s0);
            s1 = fma(Lpp[i * 17 + k + 1], x[k + 1], s1);
        }
        if (i & 1) s0 = fma(Lpp[i * 17 + i - 1], x[i - 1], s0);
        const double di = invd[i];
        x[i] = (i == j) ? di : -di * (s0 + s1);
    }
}

// Compile-time work lists of wavefront W: for every panel p the slots whose tile lies right of the panel column
// (trailing update), and for every tile row r the slots whose tile (r, q), q < min(r, PVT), is kept in registers
// after P3 (back-substitution).
template <class D, int TPW, int W, bool PIPE = false>
struct WaveLists {
    int ntrail[D::NT];
    int trail[D::NT][TPW];
    int nrow[D::NT];
    int row[D::NT][TPW];
    // PIPE: the update of panel p in two parts -- the tiles of column p + 1 (the next panel: `first`, on the critical path)
    // and everything right of it (`rest`, under the next panel's stream)
    int nfirst[D::NT];
    int first[D::NT][TPW];
    int nrest[D::NT];
    int rest[D::NT][TPW];
    constexpr WaveLists() : ntrail{}, trail{}, nrow{}, row{}, nfirst{}, first{}, nrest{}, rest{} {
        constexpr TileTab<D, PIPE> tab{};
        for (int p = 0; p < D::NT; ++p) {
            for (int q = 0; q < TPW; ++q) {
                const int t = q * D::NWAVES + W;
                if (!tab.holds(t, W)) continue;
                if (tab.tj[t] > p) trail[p][ntrail[p]++] = q;
                if (tab.tj[t] == p + 1) first[p][nfirst[p]++] = q;
                if (tab.tj[t] > p + 1) rest[p][nrest[p]++] = q;
                if (tab.ti[t] == p && tab.tj[t] < p && tab.tj[t] < D::PVT) row[p][nrow[p]++] = q;
            }
        }
    }
};

// VS_DIAG_P3 (measurement builds, with the stamps instantiation): where P3's cycles go, seen from wavefront 0 -- panel
// stream, wait at the barrier behind it, diagonal store + reloads + trailing update, wait at the barrier behind that.
// Reported by tools/gpu_phases.py in place of the P1 detail rows.
#ifdef VS_DIAG_P3
__shared__ unsigned long long vs_diag_p3[4];
#define VS_P3_MARK(i)                                                                                 \
    do {                                                                                              \
        if (DEBUG && W == 0) {                                                                        \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                             \
            if (lane == 0) vs_diag_p3[i] += now_ - p3_mark;                                           \
            p3_mark = now_;                                                                           \
        }                                                                                             \
    } while (0)
#else
#define VS_P3_MARK(i) do { } while (0)
#endif
// PIPE: acc[q] -= L_ip L_jp^T for the slots of one work list of wavefront W (KIND 1: the tiles of column PP + 1, which are
// then handed to LDS as the next panel; KIND 2: everything right of it), panel column PP in LDS.  Two tiles at a time:
// independent v_mfma_f64_16x16x4_f64 issue every 64 cycles, a dependent one every ~95 -- since the pipelined schedule put
// these chains on the critical path (first) or beside a panel stream that is no longer than they are (rest), that matters.
// The operands of the next pair are requested before the chains of the current one.
// KIND 1: the tiles of column PP + 1 (3: only its diagonal tile, 4: all but the diagonal tile), 2: everything right of it
template <class D, int TPW, int W, int PP, int KIND>
constexpr int pipe_slot(int a) {
    constexpr WaveLists<D, TPW, W, true> wl{};
    constexpr TileTab<D, true> tab{};
    if (KIND == 2) return a < wl.nrest[PP] ? wl.rest[PP][a] : -1;
    int k = 0;
    for (int b = 0; b < wl.nfirst[PP]; ++b) {
        const int t = wl.first[PP][b] * D::NWAVES + W;
        const bool dg = tab.ti[t] == tab.tj[t];
        if (KIND == 1 || (KIND == 3 && dg) || (KIND == 4 && !dg)) {
            if (k == a) return wl.first[PP][b];
            ++k;
        }
    }
    return -1;
}
template <class D, int TPW, int W, int PP, int KIND>
constexpr int pipe_count() {
    int n = 0;
    while (n < TPW && pipe_slot<D, TPW, W, PP, KIND>(n) >= 0) ++n;
    return n;
}
// the most tiles any of the wavefronts 1..3 has to update between the barrier that says "diagonal tile of column PP + 1
// ready" and the one inside the next panel stream (decides how far into the stream that barrier sits)
template <class D, int TPW, int PP>
constexpr int pipe_max_others() {
    const int n1 = pipe_count<D, TPW, 1, PP, 4>(), n2 = pipe_count<D, TPW, 2, PP, 4>(), n3 = pipe_count<D, TPW, 3, PP, 4>();
    return n1 > n2 ? (n1 > n3 ? n1 : n3) : (n2 > n3 ? n2 : n3);
}
template <class D, int TPW, int W, int PP, int KIND>
VS_DEV void pipe_update(d4 (&acc)[TPW], double* __restrict__ sM, int lrow, int crow) {
    constexpr TileTab<D, true> tab{};
    constexpr int n = pipe_count<D, TPW, W, PP, KIND>();
    double la[2][2][4], lb[2][2][4];   // [buffer][tile of the pair][k-step]
    auto request = [&](auto acst) __attribute__((always_inline)) {
        constexpr int a = decltype(acst)::value;   // first tile of the pair
        static_for<0, 2>([&](auto ucst) __attribute__((always_inline)) {
            constexpr int u = decltype(ucst)::value;
            if constexpr (a + u < n) {
                constexpr int t = pipe_slot<D, TPW, W, PP, KIND>(a + u) * D::NWAVES + W;
                const double* Lip = sM + tile_off_c<D>(tab.ti[t], PP) + lrow;
                const double* Ljp = sM + tile_off_c<D>(tab.tj[t], PP) + lrow;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) { la[(a >> 1) & 1][u][ks] = -Lip[4 * ks]; lb[(a >> 1) & 1][u][ks] = Ljp[4 * ks]; }
            }
        });
    };
    request(std::integral_constant<int, 0>{});
    static_for<0, (TPW + 1) / 2>([&](auto hcst) __attribute__((always_inline)) {
        constexpr int a = 2 * decltype(hcst)::value;
        if constexpr (a < n) {
            request(std::integral_constant<int, a + 2>{});
            constexpr int q0 = pipe_slot<D, TPW, W, PP, KIND>(a), q1 = pipe_slot<D, TPW, W, PP, KIND>(a + 1 < n ? a + 1 : a);
            constexpr int b = (a >> 1) & 1;
            if constexpr (a + 1 < n) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    acc[q0] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[b][0][ks], lb[b][0][ks], acc[q0], 0, 0, 0);
                    acc[q1] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[b][1][ks], lb[b][1][ks], acc[q1], 0, 0, 0);
                }
            } else if constexpr (KIND == 2 || KIND == 1) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    acc[q0] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[b][0][ks], lb[b][0][ks], acc[q0], 0, 0, 0);
            } else {   // (long horizons) a lone tile on the critical path: two chains of two, summed (a dependent step costs ~95 cycles, not 64)
                d4 c2 = d4{0.0, 0.0, 0.0, 0.0};
                acc[q0] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[b][0][0], lb[b][0][0], acc[q0], 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(la[b][0][1], lb[b][0][1], c2, 0, 0, 0);
                acc[q0] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[b][0][2], lb[b][0][2], acc[q0], 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(la[b][0][3], lb[b][0][3], c2, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[q0][r] += c2[r];
            }
            if constexpr (KIND != 2) {
                static_for<0, 2>([&](auto ucst) __attribute__((always_inline)) {
                    constexpr int u = decltype(ucst)::value;
                    if constexpr (a + u < n) {
                        constexpr int q = u ? q1 : q0;
                        double* T = sM + tile_off_c<D>(tab.ti[q * D::NWAVES + W], PP + 1) + crow;
#pragma unroll
                        for (int r = 0; r < 4; ++r) T[4 * r * 17] = acc[q][r];
                    }
                });
            }
        }
    });
}

// X = L_pp^-1 by the DPP rows stream on the identity (tools/gen_panel_asm.py, stream_inverse): 234 instructions, ~1.3 k cycles
// against ~2.5 k for tile_inverse (LDS broadcast + FMA pairs)
VS_DEV void tile_inverse_dpp(const double* __restrict__ Lpp, const double* __restrict__ invd, double* __restrict__ X, int lane) {
    panel_inverse_dpp(lds_addr(Lpp + (lane & 15) * 17), lds_addr(invd), lds_addr(X + (lane & 15)), lane & 15);
}

// Which of the wavefronts 1..3 inverts the diagonal tile of panel p - 1 while panel p is streamed (PIPE): the one with the
// fewest tiles in the update that runs beside it.
template <class D, int TPW>
constexpr int pipe_inverse_wave(int p) {
    constexpr WaveLists<D, TPW, 1, true> w1{};
    constexpr WaveLists<D, TPW, 2, true> w2{};
    constexpr WaveLists<D, TPW, 3, true> w3{};
    const int n1 = w1.nrest[p - 1], n2 = w2.nrest[p - 1], n3 = w3.nrest[p - 1];
    return (n3 <= n1 && n3 <= n2) ? 3 : (n2 <= n1 ? 2 : 1);
}

template <class D, int TPW, int W, bool DEBUG, bool PLDS, bool PIPE = false>
VS_DEV void cholesky_wave(const double* __restrict__ sCfg, d4 (&acc)[TPW], double* __restrict__ sM, double* __restrict__ sInvD,
                          const double* __restrict__ sGy, const double* __restrict__ sVprev, int* __restrict__ sFlags,
                          double* __restrict__ sXinv, double* __restrict__ sW, double* __restrict__ dbgL, int lane,
                          int crow, int lrow, double* sZ_) {
    constexpr TileTab<D, PIPE> tab{};
    constexpr WaveLists<D, TPW, W, PIPE> wl{};
    using S = Smem<D>;
    constexpr int PVT = D::PVT;
    constexpr int GL = D::NZ & 15;  // local row of the gradient row (row NZ) in the last tile row
    if constexpr (D::WG_PER_CU == 1) pin_tiles_agpr<TPW>(acc);
    // ---- P2: the input-cost terms, by kind of tile (compile time).  A joint diagonal tile gets the joint weights on its
    // diagonal, a tile of throttle rows x joint columns only the regularisation term of the gradient row, and only the
    // throttle x throttle tiles go through the general (branchy) input_cost_term.  (Through v13 every element of every
    // such tile went through it: 23k instructions of control flow at the 2x horizon, whose saved execution masks were what
    // pushed scalar registers into vector lanes and accumulator tiles into scratch.)
    static_for<0, TPW>([&](auto qcst) __attribute__((always_inline)) {
        constexpr int q = decltype(qcst)::value;
        constexpr int t = q * D::NWAVES + W;
        if constexpr (tab.forms(t, W)) {
            constexpr int ti = tab.ti[t], tj = tab.tj[t];
            if constexpr (ti == tj && 16 * ti + 16 <= D::NU) {
                // unit weights on the reduced joint unknowns (U^T W U / 2 = |y|^2 / 2 + |n|^2 / 2, costsVSMPC.cpp:375-381,
                // 564-571 through the joint reduction) and on the dummy unknowns behind them
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[q][r] += ((lane >> 4) + 4 * r == (lane & 15)) ? 1.0 : 0.0;
            } else if constexpr (ti >= PVT && 16 * tj + 16 <= D::NU) {
                if constexpr (ti == D::NT - 1) {   // the gradient row: Q^T b, b = w_reg W^(-1/2) q_err (costsVSMPC.cpp:586-590)
                    const int gc = 16 * tj + (lane & 15);
                    const double gv = sGy[gc % NJC];
                    const double gq = gc < D::NUY ? gv : 0.0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[q][r] += (16 * ti + (lane >> 4) + 4 * r == D::NZ) ? gq : 0.0;
                }
            } else if constexpr (ti == tj || ti >= PVT) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[q][r] += input_cost_term<D>(sCfg, sGy, sVprev, 16 * ti + (lane >> 4) + 4 * r, 16 * tj + (lane & 15));
            }
        }
    });
    if constexpr (D::WG_PER_CU == 1) pin_tiles_agpr<TPW>(acc);
    // ---- P3: tile column 0 goes to LDS; every later column is stored by the update that completes it
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int t = q * D::NWAVES + W;
        if (tab.forms(t, W) && tab.tj[t] == 0) {
            double* T = sM + tile_off_c<D>(tab.ti[t], 0) + crow;
#pragma unroll
            for (int r = 0; r < 4; ++r) T[4 * r * 17] = acc[q][r];
        }
    }
    __syncthreads();
#ifdef VS_DIAG_P3
    unsigned long long p3_mark = __builtin_amdgcn_s_memtime();
    if (DEBUG && W == 0 && lane < 4) vs_diag_p3[lane] = 0;
#endif
    // ---------------------------------------------------------------- PIPE: the pipelined schedule (kernel v27)
    // Wavefront 0 factors panel p while wavefronts 1..3 -- which hold all the tiles -- apply panel p - 1 to everything right of
    // column p.  Only the update of column p + 1 itself (`first`: one or two tiles per wavefront) stands between two streams:
    //     wavefront 0                         wavefronts 1..3
    //     stream(p): pivots 0..KB of the      first-update(p - 1) of the tiles (i, p), i > p, handed to LDS
    //     diagonal tile ...
    //     ----- s_barrier inside the stream = barrier: column p complete
    //     ... the rows below join, rest of    column p - 1's finished tiles -> registers; rest-update(p - 1): tiles (i, j), j > p,
    //     the stream                          with column p - 1;  one of them: X_(p-1)
    //     ----------------------------------- barrier: column p factored, the ring slot of column p - 1 free
    //     store the diagonal tile             first-update(p) of tile (p + 1, p + 1) with column p, handed to LDS
    //     ----------------------------------- barrier: diagonal tile of column p + 1 complete
    // Panel 0 has no update beside it: it is shared like in the plain schedule (one 64-row stream per wavefront); from
    // panel 1 on wavefront 0 carries all rows below the diagonal tile in one to three row slots.
    if constexpr (PIPE) {
        static_assert(VS_PANEL_DPP, "the pipelined schedule is built on the DPP panel streams");
        // Long horizons (one workgroup per CU; panel columns up to eleven tiles high): the diagonal tile of the next column is
        // updated by wavefront 0 itself and the other tiles arrive under the first pivots of its stream (ND, below).  Measured:
        // 1,637 -> 1,596 us per 4096 instances at the 2x horizon; at the paper horizon, where a column is at most six tiles
        // and the other wavefronts are done with them in the time wavefront 0 needs for its one, 37.4 us against 37.2.
#ifndef VS_P3_NEXTDIAG
#define VS_P3_NEXTDIAG (D::WG_PER_CU == 1)
#endif
        constexpr bool ND = VS_P3_NEXTDIAG;
        static_for<0, D::NT>([&](auto pcst) __attribute__((always_inline)) {
            constexpr int p = decltype(pcst)::value;
            if constexpr (D::WG_PER_CU == 1) pin_tiles_agpr<TPW>(acc);
            constexpr int below = D::NP - 16 * p - 16;
            constexpr int nshare0 = (below + 63) / 64;
            constexpr int SL = below <= 64 ? 1 : (below <= 128 ? 2 : 3);
            static_assert(p == 0 ? nshare0 <= D::NWAVES - 1 : below <= 192, "rows of a panel fit the streams");
            double* scratch = p <= PVT ? sXinv + p * D::TS + 64 * W : sZ_ + 64 * W;
            double diag[16];
            if constexpr (p == D::NT - 1) {
                constexpr int NPIV_LAST = D::NZ - 16 * (D::NT - 1);
                if constexpr (NPIV_LAST == 8 || NPIV_LAST == 12) {   // the generated streams (tools/gen_panel_asm.py LAST_PANEL_PIVOTS)
                    if (W == 0 && panel_last_dpp<D, NPIV_LAST>(sM, sInvD, lane) && lane == 0) sFlags[0] = 1;
                } else {                                              // any other horizon: the C++ stream
                    if (W == 0 && panel_factor<D, 1, NPIV_LAST, false, false>(sM, sInvD, p, lane, 0, diag, scratch) && lane == 0) sFlags[0] = 1;
                }
            } else if constexpr (p == 0) {
                if (W < nshare0) {
                    const int bad = panel_dpp<D, 1>(sM, sInvD, p, lane, W, diag, scratch);
                    if (W == 0 && bad && lane == 0) sFlags[0] = 1;
                }
            } else if constexpr (W == 0) {
                // the stream starts on the diagonal tile alone; the barrier that says "the rows below are complete" is the
                // s_barrier INSIDE it (behind pivot KB: far enough in for the other wavefronts' one or two -- long horizons:
                // up to four -- tiles of this column), matched by the __syncthreads() behind their first-update below
                constexpr int KB = !ND ? 0 : (SL == 3 ? 6 : (SL == 1 ? 3 : (pipe_max_others<D, TPW, p - 1>() > 2 ? 6 : 3)));
                const int bad = panel_dpp<D, SL, KB>(sM, sInvD, p, lane, 0, diag, scratch);
                if (bad && lane == 0) sFlags[0] = 1;
            }
            // the holder of tile (p + 1, p + 1) parks it in LDS (all earlier panels applied): wavefront 0 applies panel p to it
            // itself as soon as its stream has ended, no wavefront has to be waited for (not for the last panel, which is that
            // one tile: its holder updates it as before)
            auto park_next_diag = [&]() __attribute__((always_inline)) {
                if constexpr (ND && W >= 1 && p + 1 < D::NT - 1) {
                    static_for<0, TPW>([&](auto qcst) __attribute__((always_inline)) {
                        constexpr int q = decltype(qcst)::value;
                        constexpr int t = q * D::NWAVES + W;
                        if constexpr (tab.holds(t, W)) {
                            if constexpr (tab.ti[t] == p + 1 && tab.tj[t] == p + 1) {
                                double* T = sM + (S::oNextDiag - S::oM) + crow;
#pragma unroll
                                for (int r = 0; r < 4; ++r) T[4 * r * 17] = acc[q][r];
                            }
                        }
                    });
                }
            };
            if constexpr (p == 0) park_next_diag();
            if constexpr (p >= 1 && W >= 1) {
                // the finished tiles of column p - 1 come back into the registers that held them (the factor P5 reads; the
                // gradient row -> right-hand side of the back-substitution).  Here, beside the stream, not between two streams:
                // their ring slot stays intact until the barrier that ends this stream.
                if constexpr (p - 1 < PVT) {
#pragma unroll
                    for (int q = 0; q < TPW; ++q) {
                        const int t = q * D::NWAVES + W;
                        if (tab.holds(t, W) && tab.tj[t] == p - 1 && tab.ti[t] > p - 1) {
                            const double* T = sM + tile_off_c<D>(tab.ti[t], p - 1) + crow;
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[q][r] = T[4 * r * 17];
                            if (tab.ti[t] == D::NT - 1 && (lane >> 4) == (GL & 3)) sW[16 * (p - 1) + (lane & 15)] = -acc[q][GL >> 2];
                        }
                    }
                }
                pipe_update<D, TPW, W, p - 1, 2>(acc, sM, lrow, crow);   // rest-update(p - 1)
                park_next_diag();
#ifndef VS_DIAG_NO_TINV
                if constexpr (W == pipe_inverse_wave<D, TPW>(p)) {
                    if constexpr (p - 1 < S::NXT)
                        tile_inverse_dpp(sM + tile_off_c<D>(p - 1, p - 1), sInvD + 16 * (p - 1), sXinv + (p - 1) * D::TS, lane);
                    if constexpr (S::DUAL3 && p - 1 == D::PVT + 1)
                        tile_inverse_dpp(sM + tile_off_c<D>(p - 1, p - 1), sInvD + 16 * (p - 1), sM + (S::oDual3T0 - S::oM), lane);
                }
#endif
            }
            VS_P3_MARK(0);
            __syncthreads();
            VS_P3_MARK(1);
            if constexpr (p + 1 < D::NT) {
                if (W == 0 && lane < 16) {  // nobody reads tile (p, p) before the next barrier
                    double* Tpp = sM + tile_off_c<D>(p, p) + lane * 17;
#pragma unroll
                    for (int c = 0; c < 16; ++c) Tpp[c] = diag[c];
                    if (DEBUG && dbgL != nullptr) {
#pragma unroll
                        for (int c = 0; c < 16; ++c)
                            if (c <= lane) dbgL[size_t(16 * p + lane) * D::NP + 16 * p + c] = diag[c];
                    }
                }
                if constexpr (ND && p + 1 < D::NT - 1) {
                    // first-update(p).  Wavefront 0: the diagonal tile of column p + 1, from its parked copy, written where the
                    // next stream loads it (same wavefront: LDS operations stay in order, no barrier) -- two chains of two matrix
                    // instructions.  Wavefronts 1..3: the other tiles of the column, handed to LDS, then the barrier that
                    // wavefront 0 meets INSIDE its next stream, behind the first pivots of the diagonal tile.
                    if constexpr (W == 0) {
                        const double* Ljp = sM + tile_off_c<D>(p + 1, p) + lrow;
                        const double* Cn = sM + (S::oNextDiag - S::oM) + crow;
                        double lb[4];
                        d4 c, c2 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) lb[ks] = Ljp[4 * ks];
#pragma unroll
                        for (int r = 0; r < 4; ++r) c[r] = Cn[4 * r * 17];
                        c = __builtin_amdgcn_mfma_f64_16x16x4f64(-lb[0], lb[0], c, 0, 0, 0);
                        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-lb[1], lb[1], c2, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f64_16x16x4f64(-lb[2], lb[2], c, 0, 0, 0);
                        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-lb[3], lb[3], c2, 0, 0, 0);
                        double* T = sM + tile_off_c<D>(p + 1, p + 1) + crow;
#pragma unroll
                        for (int r = 0; r < 4; ++r) T[4 * r * 17] = c[r] + c2[r];
                        VS_P3_MARK(2);
                    } else {
                        pipe_update<D, TPW, W, p, 4>(acc, sM, lrow, crow);
                        __syncthreads();
                    }
                } else {
                    // short horizons, and the last panel (one tile) everywhere: the holders apply panel p to the whole column,
                    // a barrier, wavefront 0 factors it
                    if constexpr (W >= 1) pipe_update<D, TPW, W, p, 1>(acc, sM, lrow, crow);
                    VS_P3_MARK(2);
                    __syncthreads();
                    VS_P3_MARK(3);
                }
            }
        });
        return;
    }
    // (a compile-time loop: the work lists below are indexed with p in constant expressions)
    static_for<0, D::NT>([&](auto pcst) __attribute__((always_inline)) {
        constexpr int p = decltype(pcst)::value;
        if constexpr (D::WG_PER_CU == 1) pin_tiles_agpr<TPW>(acc);   // long horizons: the tiles stay in the AGPR half
        // rows under the diagonal tile and the wavefronts that share the panel (see panel_factor): one-slot streams of 48
        // rows wherever NWAVES - 1 wavefronts cover the panel, PANEL_SLOTS-slot streams for the tall panels of long horizons
        const int below = D::NP - 16 * p - 16;
        const bool one_slot = VS_PANEL_DPP || (below + 47) / 48 <= D::NWAVES - 1;
        const int rpw = VS_PANEL_DPP ? 64 : one_slot ? 48 : 64 * D::PANEL_SLOTS - 16;
        const int nshare = below <= rpw ? 1 : (below + rpw - 1) / rpw;
        static_assert((D::NP - 16 + 63) / 64 <= D::NWAVES - 1, "panel 0 leaves one wavefront for the side work");
        double diag[16];  // factored diagonal tile of a shared panel (wavefront 0, lanes 0..15), stored after the barrier
        // broadcast strip of this wavefront (64 doubles): the slot of X_p, which nobody writes before panel p + 1; the last
        // panels (one wavefront each) borrow the not-yet-used z vector
        double* sCol = p <= PVT ? sXinv + p * D::TS + 64 * W : sZ_ + 64 * W;
        static_assert(D::TS >= 64 * D::NWAVES && D::NP >= 64, "broadcast strips");
        {
            constexpr int NPIV_LAST = D::NZ - 16 * (D::NT - 1);
            if (p == D::NT - 1) {
                if constexpr (VS_PANEL_DPP && (NPIV_LAST == 8 || NPIV_LAST == 12)) {
                    if (W == 0 && panel_last_dpp<D, NPIV_LAST>(sM, sInvD, lane) && lane == 0) sFlags[0] = 1;
                } else {
                    if (W == 0 && panel_factor<D, 1, NPIV_LAST, false, PLDS>(sM, sInvD, p, lane, 0, diag, sCol) && lane == 0) sFlags[0] = 1;
                }
            } else if (W < nshare) {
                int bad;
                if constexpr (VS_PANEL_DPP) bad = panel_dpp<D>(sM, sInvD, p, lane, W, diag, sCol);
                else if (one_slot) bad = panel_factor<D, 1, 16, true, PLDS>(sM, sInvD, p, lane, W, diag, sCol);
                else bad = panel_factor<D, D::PANEL_SLOTS, 16, true, PLDS>(sM, sInvD, p, lane, W, diag, sCol);
                if (W == 0 && bad && lane == 0) sFlags[0] = 1;
            }
        }
        // a wavefront without panel rows inverts the diagonal tile finished one panel ago (its ring slot is intact
        // until the update of THIS panel hands column p+1 over): X_0..X_PVT for P5 and the dual box QP
#ifndef VS_DIAG_NO_TINV   // (measurement builds: P3 without the tile inverses; results are garbage)
        if (W == (nshare > 1 ? D::NWAVES - 1 : 1) && p >= 1 && p - 1 < S::NXT)
#else
        if (false)
#endif
            tile_inverse<D>(sM + tile_off_c<D>(p - 1, p - 1), sInvD + 16 * (p - 1), sXinv + (p - 1) * D::TS, lane);
        // three throttle tile rows: the box QP also wants the inverses of the other two corner diagonal tiles.  The second
        // one here, into the tile-shaped scratch the QP reads it from (that part of the ring is dead since the last joint
        // panel); the third one after P3 (solve_kernel).  ~4 k cycles each that used to sit at the top of the box QP.
        if constexpr (S::DUAL3) {
            if (W == (nshare > 1 ? D::NWAVES - 1 : 1) && p - 1 == D::PVT + 1)
                tile_inverse<D>(sM + tile_off_c<D>(p - 1, p - 1), sInvD + 16 * (p - 1), sM + (S::oDual3T0 - S::oM), lane);
        }
        VS_P3_MARK(0);
        __syncthreads();
        VS_P3_MARK(1);
        if (W == 0 && p < D::NT - 1 && lane < 16) {  // nobody reads tile (p, p) before the next barrier
            double* Tpp = sM + tile_off_c<D>(p, p) + lane * 17;
#pragma unroll
            for (int c = 0; c < 16; ++c) Tpp[c] = diag[c];
            if (DEBUG && dbgL != nullptr) {
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c <= lane) dbgL[size_t(16 * p + lane) * D::NP + 16 * p + c] = diag[c];
            }
        }
        if (p + 1 < D::NT) {
            // finished tiles of panel column p come back into the registers that held them (columns of the throttle
            // corner stay in LDS); the gradient row -> right-hand side y = -L^-1 g of the back-substitution.  Requested
            // first: the loads complete under the matrix-core work below.
            if (p < PVT) {
#pragma unroll
                for (int q = 0; q < TPW; ++q) {
                    const int t = q * D::NWAVES + W;
                    if (t < D::NTRI && tab.tj[t] == p && tab.ti[t] > p) {
                        const double* T = sM + tile_off_c<D>(tab.ti[t], p) + crow;
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[q][r] = T[4 * r * 17];
                        if (tab.ti[t] == D::NT - 1 && (lane >> 4) == (GL & 3)) sW[16 * p + (lane & 15)] = -acc[q][GL >> 2];
                    }
                }
            }
            // trailing update M_ij -= L_ip L_jp^T for the owned tiles right of the panel; the operands of the next
            // tile are requested before the four matrix-core instructions of the current one.  (Two tiles at a time --
            // interleaved chains, a dependent v_mfma_f64_16x16x4_f64 issues every ~95 cycles, independent ones every 64 --
            // measured no faster at either horizon: the panel streams bound P3, not these.)
            double la[2][4], lb[2][4];
            auto request = [&](auto acst) __attribute__((always_inline)) {
                constexpr int a = decltype(acst)::value;
                if constexpr (a < wl.ntrail[p]) {
                    constexpr int t = wl.trail[p][a] * D::NWAVES + W;
                    const double* Lip = sM + tile_off_c<D>(tab.ti[t], p) + lrow;
                    const double* Ljp = sM + tile_off_c<D>(tab.tj[t], p) + lrow;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) { la[a & 1][ks] = -Lip[4 * ks]; lb[a & 1][ks] = Ljp[4 * ks]; }
                }
            };
            request(std::integral_constant<int, 0>{});
            static_for<0, TPW>([&](auto acst) __attribute__((always_inline)) {
                constexpr int a = decltype(acst)::value;
                if constexpr (a < wl.ntrail[p]) {
                    request(std::integral_constant<int, a + 1>{});
                    constexpr int q = wl.trail[p][a];
                    constexpr int t = q * D::NWAVES + W;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[a & 1][ks], lb[a & 1][ks], acc[q], 0, 0, 0);
                    if constexpr (tab.tj[t] == p + 1) {  // this tile column is the next panel: hand it to LDS
                        double* T = sM + tile_off_c<D>(tab.ti[t], p + 1) + crow;
#pragma unroll
                        for (int r = 0; r < 4; ++r) T[4 * r * 17] = acc[q][r];
                    }
                }
            });
            VS_P3_MARK(2);
            __syncthreads();
            VS_P3_MARK(3);
        }
    });
}

// ------------------------------------------------------------------------------------------------
// Sum of a per-lane value over the four 16-lane rows of a wavefront (lanes l, l^16, l^32, l^48), result in every lane.
// gfx950's v_permlane16_swap / v_permlane32_swap exchange rows / halves between two registers at VALU latency
// (a ds_bpermute-based __shfl_xor costs an LDS round trip per step).
// ------------------------------------------------------------------------------------------------
VS_DEV double row_sum4(double x) {
    unsigned lo = __double2loint(x), hi = __double2hiint(x);
    auto l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double s = __hiloint2double(h2[0], l2[0]) + __hiloint2double(h2[1], l2[1]);
    lo = __double2loint(s);
    hi = __double2hiint(s);
    auto l3 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto h3 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(h3[0], l3[0]) + __hiloint2double(h3[1], l3[1]);
}

// ------------------------------------------------------------------------------------------------
// P5 for wavefront W: back-substitution L^T z = w from the REGISTER-resident factor, tile row by tile row from
// the bottom.  z_r is known (throttle rows: sZ) or formed by every wavefront redundantly and bit-identically on the
// matrix core: z_r = X_r^T d, d = w_r - sum of the wavefronts' published partial sums, as D = A B with A = X_r^T and
// every column of B = d -- the result arrives in the accumulator layout, i.e. lane (g, j) holds z[g + 4 i], exactly
// the operand layout of the owned tiles (lane (g, j) holds L[g + 4 i][j]).  Every owned tile (r, q) then adds
// L_rq^T z_r to this wavefront's partial sum u_q, kept in registers; the partial sums of column r - 1 are published
// before the barrier that ends step r (fixed summation order -> deterministic).  One workgroup barrier per tile row.
// ------------------------------------------------------------------------------------------------
template <class D, int TPW, int W, bool PIPE = false>
VS_DEV void backsub_wave(const d4 (&acc)[TPW], const double* __restrict__ sW, double* __restrict__ sZ,
                         const double* __restrict__ sXinv, double* __restrict__ sU, int lane) {
    constexpr WaveLists<D, TPW, W, PIPE> wl{};
    constexpr TileTab<D, PIPE> tab{};
    constexpr int PVT = D::PVT;
    double* myU = sU + W * D::NP;
    const int j = lane & 15, g4 = lane >> 4;
    double up[PVT];
#pragma unroll
    for (int q = 0; q < PVT; ++q) up[q] = 0.0;
    double xop[4];  // A operand of the next joint tile row: X_r[g4 + 4 ks][j], requested one step ahead
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) xop[ks] = sXinv[(PVT - 1) * D::TS + (g4 + 4 * ks) * 17 + j];
#pragma unroll
    for (int r = D::NT - 1; r >= 0; --r) {
        double zz[4];
        if (r < PVT) {
            double dop[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int row = 16 * r + g4 + 4 * ks;
                double usum = sU[row];
#pragma unroll
                for (int w = 1; w < D::NWAVES; ++w) usum += sU[w * D::NP + row];
                dop[ks] = sW[row] - usum;
            }
            // two accumulators, summed: two dependent pairs instead of a chain of four (~95 cycles per dependent step)
            d4 zt = d4{0.0, 0.0, 0.0, 0.0}, zu = d4{0.0, 0.0, 0.0, 0.0};
            zt = __builtin_amdgcn_mfma_f64_16x16x4f64(xop[0], dop[0], zt, 0, 0, 0);
            zu = __builtin_amdgcn_mfma_f64_16x16x4f64(xop[1], dop[1], zu, 0, 0, 0);
            zt = __builtin_amdgcn_mfma_f64_16x16x4f64(xop[2], dop[2], zt, 0, 0, 0);
            zu = __builtin_amdgcn_mfma_f64_16x16x4f64(xop[3], dop[3], zu, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) zt[i] += zu[i];
            if (r > 0) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) xop[ks] = sXinv[(r - 1) * D::TS + (g4 + 4 * ks) * 17 + j];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) zz[i] = zt[i];
            if (W == (PIPE ? 1 : 0) && j == 0) {   // (PIPE: wavefront 0 is not here, see p5_wave0_jets)
#pragma unroll
                for (int i = 0; i < 4; ++i) sZ[16 * r + g4 + 4 * i] = zz[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) zz[i] = sZ[16 * r + g4 + 4 * i];
        }
        if (r > 0) {
#pragma unroll
            for (int a = 0; a < TPW; ++a) {
                if (a < wl.nrow[r]) {
                    const int q = wl.row[r][a];
                    const int t = q * D::NWAVES + W;
                    double part = acc[q][0] * zz[0];
#pragma unroll
                    for (int i = 1; i < 4; ++i) part = fma(acc[q][i], zz[i], part);
                    up[tab.tj[t]] += part;   // per 16-lane row; the four rows are summed once, when the column is published
                }
            }
            if (r - 1 < PVT) {
                const double usum = row_sum4(up[r - 1]);
                if (lane < 16) myU[16 * (r - 1) + j] = usum;
                __syncthreads();   // (a step that publishes nothing -- the corner tile rows above the first joint row -- needs none:
                                   // PVT barriers in all, which p5_wave0_jets matches)
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// P5, wavefront 0 of the pipelined schedule (kernel v29).  It holds no tile, so the back-substitution has nothing for it to do
// but meet its barriers -- and the first link of P6's cascade, the jets, depends on the throttles only, which are final
// since P4: their input terms and their whole two-state recursion (systemDynamicsVSMPC.cpp:384-429) run here, beside P5,
// CHJ stages per barrier interval.  P6 then starts at the momenta, one pipeline step shorter.  (Wavefront 0's slot of the
// partial sums is zeroed once; wavefront 1 stores z.)
// ------------------------------------------------------------------------------------------------
template <class D>
VS_DEV void p5_wave0_jets(double* __restrict__ smem, int lane) {
    using S = Smem<D>;
    // Stages per barrier interval.  P5 has one barrier per joint tile row (PVT of them; the corner tile rows in front publish
    // nothing and have none): a chunk behind every barrier but the last, beside the other wavefronts' joint-row steps (~0.9 k
    // cycles each).  Measured (when the corner rows still had barriers): 3 stages in every interval P5 6.1 k cycles, 6 in the
    // first three 6.8 k, 9 in the first two 7.2 k (5.5 k without the jets); chunks beside the short corner-row steps delay them.
    constexpr int BI0 = 0;                                 // first interval that gets a chunk
    constexpr int NIV = D::PVT - 1;                        // intervals with a chunk
#ifndef VS_P5_CHJ
#define VS_P5_CHJ ((D::N + NIV - 1) / NIV)
#endif
    constexpr int CHJ = VS_P5_CHJ, NCJ = (D::N + CHJ - 1) / CHJ;
    static_assert(NIV >= 1 && NCJ <= NIV, "the jets fit the barrier intervals of P5");
    const double* sIn = smem + S::oIn;
    const double* sA = smem + S::oA;
    const double* sBt = smem + S::oBt;
    const double* sC = smem + S::oC;
    const double* sZ = smem + S::oZ;
    const double* sDt = smem + S::oDt;
    double* sU = smem + S::oU;
    double* sX = smem + S::oX;
    for (int i = lane; i < D::NP; i += 64) sU[i] = 0.0;
    const int jl = lane < NTH ? lane : 0;     // (lanes >= NTH shadow jet 0 and store nothing)
    const double jon = sA[(12 + jl) * NX + 16 + jl], ja = sA[(16 + jl) * NX + 12 + jl], jb = sA[(16 + jl) * NX + 16 + jl];
    const double c12 = sC[12 + jl], c16 = sC[16 + jl];
    double bt12[NTH], bt16[NTH];
#pragma unroll
    for (int c = 0; c < NTH; ++c) { bt12[c] = sBt[(12 + jl) * NTH + c]; bt16[c] = sBt[(16 + jl) * NTH + c]; }
    double jT = sIn[VSMPC_IN_X0 + 12 + jl], jTd = sIn[VSMPC_IN_X0 + 16 + jl];
    double fa = 0.0, fb = 0.0;   // input terms of the current throttle block
    if (lane < NTH) {
        sX[12 + lane] = jT;
        sX[16 + lane] = jTd;
    }
    static_for<0, D::PVT>([&](auto bcst) __attribute__((always_inline)) {
        constexpr int bi = decltype(bcst)::value;
        __syncthreads();
        if constexpr (bi >= BI0 && bi - BI0 < NCJ) {
            // input terms of the jet rows in place: f = Bt v_{tb(k)} + c (the joints do not reach these rows); v: uniform
            // addresses (LDS broadcasts), shared by the stages of a throttle block
            static_for<0, CHJ>([&](auto ucst) __attribute__((always_inline)) {
                constexpr int k = (bi - BI0) * CHJ + decltype(ucst)::value;
                if constexpr (k < D::N) {
                    constexpr int tb = throttle_block_of_stage<D>(k);
                    constexpr int tb_prev = k == 0 ? -1 : throttle_block_of_stage<D>(k == 0 ? 0 : k - 1);
                    if constexpr (tb != tb_prev) {   // the input term changes with the throttle block only
                        constexpr int vq = tb == 0 ? D::NV - 4 : 4 * (tb - 1);   // internal offset of reference block tb
                        fa = c12;
                        fb = c16;
#pragma unroll
                        for (int c = 0; c < NTH; ++c) {
                            const double vc = sZ[D::NU + vq + c];
                            fa = fma(bt12[c], vc, fa);
                            fb = fma(bt16[c], vc, fb);
                        }
                    }
                    const double dt = sDt[k];
                    const double dT = fma(jon, jTd, fa);
                    const double dTd = fma(ja, jT, fma(jb, jTd, fb));
                    jT = fma(dt, dT, jT);
                    jTd = fma(dt, dTd, jTd);
                    if (lane < NTH) {
                        sX[NX * (k + 1) + 12 + lane] = jT;
                        sX[NX * (k + 1) + 16 + lane] = jTd;
                    }
                }
            });
        }
    });
}

// ------------------------------------------------------------------------------------------------
// K x K symmetric positive definite system (P_AA mu = rhs_A of the dual box QP, S_FF v_F = b_F of the primal), K <= 12, all lanes redundantly on
// wave-uniform values (symmetric elimination on the lower triangle).  sP[b * NVS + i] = P[i][b]; the K set bits of
// `mask` are the active indices; lane idx[q] returns mu_q, every other lane 0.
template <int K, int NVS>
VS_DEV double small_spd_solve(const double* __restrict__ sP, unsigned long long mask, double rhs, int lane, int& bad) {
    int idx[K];
#pragma unroll
    for (int q = 0; q < K; ++q) {
        idx[q] = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
    }
    double A[K][K], d[K];
#pragma unroll
    for (int q = 0; q < K; ++q) {
        d[q] = readlane_f64(rhs, idx[q]);
#pragma unroll
        for (int c = 0; c <= q; ++c) A[q][c] = sP[idx[c] * NVS + idx[q]];  // uniform address: LDS broadcast
    }
    double ip[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        bad |= !(A[j][j] > 0.0);
        ip[j] = fast_rcp(A[j][j]);
#pragma unroll
        for (int i = j + 1; i < K; ++i) {
            const double f = A[i][j] * ip[j];
            d[i] = fma(-f, d[j], d[i]);
#pragma unroll
            for (int c = j + 1; c <= i; ++c) A[i][c] = fma(-f, A[c][j], A[i][c]);
        }
    }
    double x[K], mu = 0.0;
#pragma unroll
    for (int j = K - 1; j >= 0; --j) {
        double t = d[j];
#pragma unroll
        for (int c = j + 1; c < K; ++c) t = fma(-A[c][j], x[c], t);
        x[j] = t * ip[j];
        mu = (lane == idx[j]) ? x[j] : mu;
    }
    return mu;
}

// size dispatch for small_spd_solve (one straight-line instantiation per size)
constexpr int SMALL_SOLVE_MAX = 6;
// block principal pivoting: non-improving block steps tolerated before the least-index fallback (the oracle's value)
constexpr int AS_PATIENCE = 10;
constexpr int AS_MAX_ITER = 64;   // active-set iteration cap (status MAX_ITER beyond)
template <int NVS, int K = SMALL_SOLVE_MAX>
VS_DEV double small_spd_solve_n(int k, const double* __restrict__ sP, unsigned long long mask, double rhs, int lane, int& bad) {
    if constexpr (K == 1) {
        return small_spd_solve<1, NVS>(sP, mask, rhs, lane, bad);
    } else {
        if (k == K) return small_spd_solve<K, NVS>(sP, mask, rhs, lane, bad);
        return small_spd_solve_n<NVS, K - 1>(k, sP, mask, rhs, lane, bad);
    }
}

// s = L22 (L^-1 g)_v, whose largest entry scales the release tolerance of the box QP (row NZ of the factor holds L^-1 g)
template <class D>
VS_DEV void schur_rhs(const double* __restrict__ Lb, double* __restrict__ sSvec, int lane) {
    constexpr int PV = D::NU >> 4;
    constexpr int GR = D::NZ - 16 * (PV + 1);  // local row of NZ in tile row PV+1
    const int r = lane < D::NV ? lane : D::NV - 1;
    const double* L76 = Lb + tile_off<D>(PV + 1, PV);
    const double* L77 = Lb + tile_off<D>(PV + 1, PV + 1);
    const double* rowp = Lb + tile_off<D>(PV + (r >> 4), PV) + (r & 15) * 17;  // L22[r][c] = rowp[(c>>4)*TS + (c&15)]
    double sr = 0.0;
#pragma unroll
    for (int c = 0; c < D::NV; ++c) {
        // tile (PV, PV+1) does not exist: that load stays inside tile row PV+1 and is masked out
        const double lrc = rowp[((c >> 4) && (r >> 4)) ? D::TS + (c & 15) : (c & 15)];
        const double ellc = c < 16 ? L76[GR * 17 + c] : L77[GR * 17 + (c - 16)];
        sr = fma(c <= r ? lrc : 0.0, ellc, sr);
    }
    if (lane < D::NV) sSvec[r] = sr;
}

// ------------------------------------------------------------------------------------------------
// Accessors of X = L22^-1 (lower triangular, NV x NV) for the dual box QP.  Entries are re-read from LDS where they are
// used instead of held in 2 NV registers: with the accumulator tiles live through P4 the box QP must stay small in
// registers, or tiles get spilled for EVERY instance.
//   XTiles  two throttle tile rows (the paper horizon): rows 0..15 are the tile X6 = inverse of the first throttle
//           diagonal tile (formed in P3), rows 16.. are formed at the top of the box QP (sXr[a * NVS + j] = X[16 + a][j])
//   XDense  three throttle tile rows: sXd[j * NVS + i] = X[j][i], zero above the diagonal
// col(j, r, n) = X[j][r] restricted to the rows j < n of N; pcol(b, r, n) = P[r][b] = sum_{j < n} X[j][r] X[j][b].
// ------------------------------------------------------------------------------------------------
template <class D>
struct XTiles {
    static constexpr int NVS = D::NV + 1, NR2 = D::NV - 16;
    static constexpr int KMAX = SMALL_SOLVE_MAX;   // largest system solved in registers (the accumulator tiles are in VGPRs here)
    static constexpr int KMID = 16;                // compact row-per-lane solver for 7..16 active bounds (mid_spd_solve): it fits
                                                   // beside the accumulator tiles without a spilled register
    const double* X6;
    const double* sXr;
    VS_DEV double col(int j, int r, int n) const {
        const double t = j < 16 ? X6[j * 17 + (r & 15)] : sXr[(j - 16) * NVS + r];
        return (j < 16 ? r < 16 : j < n) ? t : 0.0;
    }
    VS_DEV double pcol(int b, int r, int n) const {
        double p0 = 0.0, p1 = 0.0;
        if (b < 16) {  // X[j][b] = 0 for j < 16 <= b
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                p0 = fma(col(j, r, n), X6[j * 17 + b], p0);            // uniform addresses: LDS broadcasts
                p1 = fma(col(j + 1, r, n), X6[(j + 1) * 17 + b], p1);
            }
        }
#pragma unroll
        for (int a2 = 0; a2 < NR2; ++a2) p0 = fma(col(16 + a2, r, n), sXr[a2 * NVS + b], p0);
        return p0 + p1;
    }
};
template <class D>
struct XDense {
    static constexpr int NVS = D::NV + 1;
    static constexpr int KMAX = 10;                // accumulator tiles live in AGPRs at these horizons: room for 10 x 10
    static constexpr int KMID = VS_KMID;           // and for the compact row-per-lane solver up to KMID x KMID (mid_spd_solve)
    const double* sXd;
    VS_DEV double col(int j, int r, int n) const { return j < n ? sXd[j * NVS + r] : 0.0; }
    VS_DEV double pcol(int b, int r, int n) const {
        double p0 = 0.0, p1 = 0.0;
        int j = b;                                   // X[j][b] = 0 for j < b
        for (; j + 1 < n; j += 2) {
            p0 = fma(sXd[j * NVS + r], sXd[j * NVS + b], p0);
            p1 = fma(sXd[(j + 1) * NVS + r], sXd[(j + 1) * NVS + b], p1);
        }
        if (j < n) p0 = fma(sXd[j * NVS + r], sXd[j * NVS + b], p0);
        return p0 + p1;
    }
};

// ------------------------------------------------------------------------------------------------
// KMAX < K <= KMID active bounds: P_AA mu = rhs_A in registers, rows where they are (lane r = throttle r keeps
// a[q] = P[r][idx_q] for the K active indices idx_0 < idx_1 < ...: compact COLUMNS, scattered rows), Gaussian elimination
// without pivoting (SPD) with the pivot rows broadcast by v_readlane from lane idx_j (a scalar).  ~5 instructions per
// update against ~12 for the elimination on an LDS copy (three LDS operations per update) and ~2.8 k instructions for one
// iteration of the primal form on all 44 throttles, which is what instances with more than 16 violated bounds ran
// before.  Lane idx_q returns mu_q, every other lane 0.
// ------------------------------------------------------------------------------------------------
template <int KMID, int NVS>
VS_DEV double mid_spd_solve(int ka, const double* __restrict__ sP, unsigned long long Amask, double bb, int lane, bool isA,
                            int& bad) {
    int idx[KMID];
    {
        unsigned long long m = Amask;
#pragma unroll
        for (int q = 0; q < KMID; ++q) {
            idx[q] = m ? __ffsll((long long)m) - 1 : 0;   // (wave uniform: scalar registers)
            m &= m - 1;
        }
    }
    const int rank = __popcll(Amask & ((1ull << lane) - 1ull));   // compact position of this lane's throttle
    double a[KMID];
#pragma unroll
    for (int q = 0; q < KMID; ++q) a[q] = sP[idx[q] * NVS + lane];   // P[lane][idx_q] (symmetric); garbage beyond ka, unused
#pragma unroll
    for (int j = 0; j < KMID; ++j) {
        if (j < ka) {   // (guards, not `break`: an early exit keeps the loops rolled and puts a[] in scratch -- measured 2.3x slower)
            const double piv = readlane_f64(a[j], idx[j]);
            bad |= !(piv > 0.0);
            const double ip = fast_rcp(piv);
            const double bj = readlane_f64(bb, idx[j]);
            const double f = (isA && rank > j) ? a[j] * ip : 0.0;
            bb = fma(-f, bj, bb);
#pragma unroll
            for (int c = j + 1; c < KMID; ++c) {
                if (c < ka) {
                    const double pc = readlane_f64(a[c], idx[j]);
                    a[c] = fma(-f, pc, a[c]);
                }
            }
        }
    }
    double mu = 0.0;
#pragma unroll
    for (int j = KMID - 1; j >= 0; --j) {
        if (j < ka) {
            const double xj = readlane_f64(bb, idx[j]) * fast_rcp(readlane_f64(a[j], idx[j]));
            mu = (lane == idx[j]) ? xj : mu;
            bb = (isA && rank < j) ? fma(-a[j], xj, bb) : bb;
        }
    }
    return mu;
}

// ------------------------------------------------------------------------------------------------
// Dual active-set iteration of the box QP by ONE wavefront (lane = throttle).  With N = the throttles that are not
// pinned by the hold, P = S_NN^-1 = X^T X and v_u = the sweep's solution (sZ), fixing the set A at its bounds b_A gives
// mu = P_AA^-1 (v_u,A - b_A),  v_N = v_u,N - P[:,A] mu,  gradient_A = -mu.  Only the columns of P some active set needs
// are ever formed; the |A| x |A| system is tiny for the usual one to three saturated throttles.  The sequence of active
// sets is exactly the block-pivoting sequence of the primal form.  sSvec[0] = max |s| (release tolerance).
// ------------------------------------------------------------------------------------------------
template <class D, class XA>
VS_DEV void dual_active_set(const XA& xa, bool hold, int lane, double* __restrict__ sSv, double* __restrict__ sQP,
                            const double* __restrict__ sSvec, const double* __restrict__ sVprev,
                            const double* __restrict__ sCfg, double* __restrict__ sZ, int* __restrict__ sFlags,
                            unsigned long long have0 = 0ull) {
    constexpr int NVS = D::NV + 1;          // row stride of the LDS work arrays
    double* sP = sSv;                       // sP[b * NVS + i] = P[i][b] for the columns b formed so far (have0: on entry)
    double* sK = sQP;                       // working copy of P_AA
    const int r = lane < D::NV ? lane : D::NV - 1;  // lanes >= NV shadow the last row (results unused)
    const bool valid = lane < D::NV;
    const bool fixed = valid && hold && (r >= D::NV - 4);  // v0 is the trailing block
    const int n = hold ? D::NV - 4 : D::NV;
    const bool inN = valid && r < n;
    const double lo = fixed ? sVprev[r & 3] : sCfg[CFG_VMIN];    // constraintsVSMPC.cpp:351-364
    const double hi = fixed ? sVprev[r & 3] : sCfg[CFG_VMAX];
    const double gtol = 1e-10 * (1.0 + sSvec[0]);   // sSvec[0] = max |s| (see above)
    const double vu = sZ[D::NU + r];
    // Iteration 1 of the block-pivoting scheme is the solve with only the hold pin enforced: that is the
    // backward sweep that just ran.  Apply its flips here; nothing is at a bound yet, so only primal
    // violations can occur.
    int state = 0;  // 0 free, -1 at lower, +1 at upper (pinned throttles are outside N altogether)
    double v = vu;
    int best, patience = AS_PATIENCE, status = VSMPC_STATUS_MAX_ITER, iters = 1, bad = 0;
    {
        const double tolv = 1e-12 * (1.0 + fabs(v));
        const bool vlo = inN && (v < lo - tolv);
        const bool vhi = inN && (v > hi + tolv);
        best = __popcll(__ballot(vlo || vhi));
        if (vlo || vhi) state = vlo ? -1 : 1;
    }
    unsigned long long have = have0;
    for (int it = 1; it < AS_MAX_ITER; ++it) {
        iters = it + 1;
        const bool isA = inN && state != 0;
        const unsigned long long Amask = __ballot(isA);
        // columns of P for the newly active throttles: P[i][b] = sum_{j < n} X[j][i] X[j][b]
        unsigned long long need = Amask & ~have;
        have |= need;
        while (need) {
            const int b = __ffsll((long long)need) - 1;
            need &= need - 1;
            const double pb = xa.pcol(b, r, n);
            if (valid) sP[b * NVS + r] = pb;
        }
        double bb = isA ? vu - (state < 0 ? lo : hi) : 0.0;  // right-hand side v_u,A - b_A
        double mu = 0.0;
        const int ka = __popcll(Amask);
        if (ka == 0) {
            // every bound was released again: v = v_u, no multipliers
        } else if (ka <= XA::KMAX) {
            // few active bounds: solved redundantly in every lane on wave-uniform values
            mu = small_spd_solve_n<D::NV + 1, XA::KMAX>(ka, sP, Amask, bb, lane, bad);
        } else if (XA::KMID > 0 && ka <= XA::KMID) {
            if constexpr (XA::KMID > 0) mu = mid_spd_solve<XA::KMID, D::NV + 1>(ka, sP, Amask, bb, lane, isA, bad);
        } else {
            // K = P_AA (working copy); Gaussian elimination without pivoting (SPD) over the active indices
            if (isA) {
                unsigned long long m = Amask;
                while (m) {
                    const int c = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    sK[r * NVS + c] = sP[c * NVS + r];
                }
            }
            for (unsigned long long pm = Amask; pm; pm &= pm - 1) {
                const int j = __ffsll((long long)pm) - 1;
                const double piv = sK[j * NVS + j];
                bad |= !(piv > 0.0);
                const double bj = readlane_f64(bb, j);
                if (isA && lane > j) {
                    const double f = sK[r * NVS + j] * fast_rcp(piv);
                    bb -= f * bj;
                    for (unsigned long long m = pm & (pm - 1); m; m &= m - 1) {
                        const int c = __ffsll((long long)m) - 1;
                        sK[r * NVS + c] -= f * sK[j * NVS + c];
                    }
                }
            }
            for (unsigned long long pm = Amask; pm;) {
                const int j = 63 - __clzll((long long)pm);
                pm &= ~(1ull << j);
                const double xj = readlane_f64(bb, j) * fast_rcp(sK[j * NVS + j]);
                if (lane == j) mu = xj;
                if (isA && lane < j) bb -= sK[r * NVS + j] * xj;
            }
        }
        if (bad) { status = VSMPC_STATUS_NUMERICAL; break; }
        // v_N = v_u,N - P[:,A] mu
        v = vu;
        for (unsigned long long m = Amask; m; m &= m - 1) {
            const int b = __ffsll((long long)m) - 1;
            const double mub = readlane_f64(mu, b);
            if (inN) v -= sP[b * NVS + r] * mub;
        }
        const double grad = -mu;  // gradient of the QP at the throttles that sit on a bound
        const double tolv = 1e-12 * (1.0 + fabs(v));
        const bool isF = inN && state == 0;
        const bool vlo = isF && (v < lo - tolv);
        const bool vhi = isF && (v > hi + tolv);
        const bool rlo = isA && state == -1 && grad < -gtol;
        const bool rhi = isA && state == 1 && grad > gtol;
        const bool inf = vlo || vhi || rlo || rhi;
        const unsigned long long imask = __ballot(inf);
        const int ninf = __popcll(imask);
        if (ninf == 0) { status = VSMPC_STATUS_SOLVED; break; }
        bool pick = inf;
        if (ninf < best) { best = ninf; patience = AS_PATIENCE; }
        else if (patience > 0) { --patience; }
        else { pick = inf && (lane == 63 - __clzll(imask)); }  // least-index fallback (largest index)
        if (pick) state = vlo ? -1 : (vhi ? 1 : 0);
    }
    if (valid) {
        v = fixed ? lo : (state < 0 ? lo : (state > 0 ? hi : v));  // bound variables sit exactly on their bound
        sZ[D::NU + lane] = v;
    }
    if (lane == 0) { sFlags[1] = status; sFlags[2] = iters; }
}

// ------------------------------------------------------------------------------------------------
// P4b: box QP on the throttles (constraintsVSMPC.cpp:338-365), entered only by instances whose pins-only solution violates
// a bound.  Kept small in registers (K x K systems up to 6 x 6 in registers, columns of X re-read from LDS): the
// accumulator tiles are live across it, and what it cannot hold gets spilled for every instance.  (Out of line as a
// real call it costs the slowest instance of a launch ~4 us in saved / restored registers.)  Few saturated throttles (the usual case): dual
// form, cost grows with the number of active bounds; many: primal form on the Schur complement, cost grows with the
// number of free throttles.  Called by all wavefronts (it contains workgroup barriers); result in sZ[NU..NZ), sFlags.
// ------------------------------------------------------------------------------------------------
template <class D>
VS_DEV void box_qp(int n_violated, bool hold, int wave) {
    using S = Smem<D>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sVprev = smem + S::oVprev;
    double* sInvD = smem + S::oInvD;
    double* sZ = smem + S::oZ;
    double* sSv = smem + S::oSv;
    double* sSvec = smem + S::oSvec;
    double* sCfg = smem + S::oCfg;
    int* sFlags = reinterpret_cast<int*>(smem + S::oFlags);
    double* Lb = smem + S::oM;
    double* sXinv = smem + S::oXinv;
    double* sQP = smem + S::oQP;
    // (not threadIdx.x: that would keep the work-item id register alive -- in scratch -- from the first instruction to here)
    const int lane = fresh_lane(), tid = (wave << 6) | lane;
    constexpr bool DUALQP = S::DUALQP;
    constexpr int PV = D::PVT;
    constexpr int DUAL_MAX_ACTIVE = 16;   // (10 before the register solver for mid-size active sets: take-off batch -1.4 %)
    constexpr bool DUAL3 = S::DUAL3;
#ifndef VS_DUAL3_MAX
#define VS_DUAL3_MAX 24
#endif
    const bool few = n_violated <= (DUAL3 ? VS_DUAL3_MAX : DUAL_MAX_ACTIVE);   // few saturated throttles: dual form
    (void)sQP; (void)sXinv; (void)sInvD;
      if (few && DUALQP) {
       if constexpr (DUALQP) {
        // ---- box QP on the throttles, dual form.  With N = the throttles that are not pinned by the hold, P = S_NN^-1
        // (S = L22 L22^T, so the factor of S_NN is the leading block of L22) and v_u = the sweep's solution, fixing the
        // set A at its bounds b_A gives  mu = P_AA^-1 (v_u,A - b_A),  v_N = v_u,N - P[:,A] mu,  gradient_A = -mu.
        // P = X^T X with X = L22^-1: rows 0..15 of X are the inverse of the first throttle diagonal tile (formed by an
        // idle wavefront during P3), the remaining rows are formed here; a column of P then is 24 multiply-adds per
        // lane with no chain, and only the columns some active set needs are ever formed.  The |A| x |A| system is
        // tiny for the usual one to three saturated throttles.  The sequence of active sets is exactly the
        // block-pivoting sequence of the primal form.
        {
            // second tile row of X by all wavefronts:  [X76 | X77] = [-X77 (L76 X66) | L77^-1]
            constexpr int NVS = D::NV + 1, NR2 = D::NV - 16;
            double* sXr = sQP + D::NV * NVS;                              // sXr[a * NVS + j] = X[16 + a][j]
            double* sT = sQP;                                             // T = L76 X66, NR2 x 16 (dead before sK is used)
            const double* X6 = sXinv + PV * D::TS;
            const double* L76 = Lb + tile_off<D>(PV + 1, PV);
            const double* L77 = Lb + tile_off<D>(PV + 1, PV + 1);
            // threads of the columns of X77: behind the 16 NR2 threads of T where that leaves wavefront 3 alone (it forms the
            // right-hand side meanwhile), else the upper lanes of wavefront 3 (throttle blocks of more than 24: both run there)
            constexpr int XT0 = 17 * NR2 <= 192 ? 16 * NR2 : 224;
            static_assert(XT0 + NR2 <= D::BLOCK && D::NV <= 32, "threads of the X77 columns");
            if (tid < 16 * NR2) {
                const int a2 = tid >> 4, j = tid & 15;
                double t0 = 0.0, t1 = 0.0;
#pragma unroll
                for (int k = 0; k < 16; k += 2) {                         // X66[k][j] = 0 for k < j (stored zeros)
                    t0 = fma(L76[a2 * 17 + k], X6[k * 17 + j], t0);
                    t1 = fma(L76[a2 * 17 + k + 1], X6[(k + 1) * 17 + j], t1);
                }
                sT[a2 * 16 + j] = t0 + t1;
            } else if (tid >= XT0 && tid < XT0 + NR2) {
                const int c = tid - XT0;                                  // column c of X77 = L77^-1
                double x[NR2];
#pragma unroll
                for (int i = 0; i < NR2; ++i) {
                    double sum = 0.0;
#pragma unroll
                    for (int k = 0; k < i; ++k) sum = fma(L77[i * 17 + k], (k >= c) ? x[k] : 0.0, sum);
                    const double di = sInvD[D::NU + 16 + i];
                    x[i] = (i == c) ? di : ((i > c) ? -di * sum : 0.0);
                    sXr[i * NVS + 16 + c] = x[i];
                }
            } else if (wave == 3) {
                schur_rhs<D>(Lb, sSvec, lane);
            }
            __syncthreads();
            if (wave == 3) {  // max |s| while wavefronts 0..1 finish X (keeps it off wavefront 0's path)
                double gm = 0.0;
#pragma unroll
                for (int c = 0; c < D::NV; ++c) gm = fmax(gm, fabs(sSvec[c]));  // uniform addresses: LDS broadcasts
                if (lane == 0) sSvec[0] = gm;   // every lane of this wavefront has read sSvec[0] (in-order LDS)
            }
            if (tid < 16 * NR2) {
                const int a2 = tid >> 4, j = tid & 15;
                double t = 0.0;
#pragma unroll
                for (int b2 = 0; b2 < NR2; ++b2) t = fma(sXr[a2 * NVS + 16 + b2], sT[b2 * 16 + j], t);  // X77[a][b] = 0, b > a
                sXr[a2 * NVS + j] = -t;
            }
            __syncthreads();
        }
        static_assert(D::NU % 16 == 0 && D::NV > 16 && D::NV <= 32, "throttle block: tile aligned, two tile rows");
        const XTiles<D> xa{sXinv + PV * D::TS, sQP + D::NV * (D::NV + 1)};   // rows 16.. of X behind the copy of P_AA
        // Many violated bounds (the take-off instances enter with ten to sixteen): all columns of P up front, by all four
        // wavefronts -- 2.25 of them per thread, ~1 k cycles -- instead of one by one in the wavefront that iterates (~0.43 k each:
        // profiles/r04_v27_qp_dist_paper.txt, 24.6 k cycles for a two-iteration solve).  Same expression, same sums.
        const bool all_cols = n_violated > 4;   // workgroup-uniform
        if (all_cols) {
            const int n = hold ? D::NV - 4 : D::NV;
            for (int e = tid; e < D::NV * D::NV; e += D::BLOCK) {
                const int b = e / D::NV, r = e - b * D::NV;
                sSv[b * (D::NV + 1) + r] = xa.pcol(b, r, n);
            }
            __syncthreads();
        }
        if (wave == 0)
            dual_active_set<D>(xa, hold, lane, sSv, sQP, sSvec, sVprev, sCfg, sZ, sFlags, all_cols ? ((1ull << D::NV) - 1ull) : 0ull);
       }
      } else if (few && DUAL3) {
       if constexpr (DUAL3) {
        // ---- dual form, three throttle tile rows: X = L22^-1 assembled dense in LDS from tile products,
        //   X_ii = L_ii^-1 (the first one from P3, the other two here),  X10 = -X1 (L10 X0),  X21 = -X2 (L21 X1),
        //   X20 = -X2 (L20 X0 + L21 X10)  (one round on the matrix cores, below)
        constexpr int NVS = D::NV + 1, R2 = D::NV - 32;        // throttle rows in the last tile row
        double* sXd = sQP + D::NV * NVS;                        // sXd[j * NVS + i] = X[j][i]
        double* sT0 = smem + S::oDual3T0;                       // X1 on entry (from P3); L10 X0, later L20 X0 + L21 X10
        double* sT1 = smem + S::oDual3T1;                       // X2 on entry (formed beside the sweep); L21 X1
        static_assert(S::oDual3T0 == S::oQP + 2 * D::NV * NVS, "scratch tiles behind K and X");
        const double* L10 = Lb + tile_off<D>(PV + 1, PV);
        const double* L20 = Lb + tile_off<D>(PV + 2, PV);
        const double* L21 = Lb + tile_off<D>(PV + 2, PV + 1);
        const double* X0 = sXinv + PV * D::TS;                  // from P3
        auto xd = [&](int blk_r, int blk_c) { return sXd + (16 * blk_r) * NVS + 16 * blk_c; };   // block of X, row stride NVS
        for (int e = tid; e < D::NV * NVS; e += D::BLOCK) sXd[e] = 0.0;
        if (tid < 4 * D::NV) {   // s = L22 (L^-1 g)_v (its largest entry scales the release tolerance): four threads per row,
            const int i = tid >> 2, part = tid & 3;   // every fourth term each -- one thread per row was a 44-step serial loop
            double sum = 0.0;                         // (4.4 k cycles) in front of the first barrier
            for (int k = part; k <= i; k += 4)
                sum += Lb[lower_at<D>(D::NU + i, D::NU + k)] * Lb[lower_at<D>(D::NZ, D::NU + k)];
            sSv[tid] = sum;                           // partial sums: P's array is not in use before round 6
        }
        __syncthreads();
        // ONE round for the off-diagonal blocks of X, on the matrix cores, each chain in one wavefront: an accumulator
        // (lane (g, n), register r = entry [g + 4 r][n]) IS the B operand of the next product (k-step r), so a chain of
        // products needs no LDS round trip and no barrier.  Wavefront 0: T10 = L10 X0, X10 = -X1 T10, T20 = L20 X0 + L21 X10,
        // X20 = -X2 T20; wavefront 1: T21 = L21 X1, X21 = -X2 T21; wavefront 2 copies the diagonal blocks; wavefront 3 forms
        // max |s|.  (Through v22: five barrier-separated rounds of 16-term dot products, one entry per thread: the set-up
        // of the box QP cost 14.5 k cycles.)  X2 is R2 x R2: its rows and columns beyond R2 do not exist (masked operands).
        {
            const int g = lane >> 4, n = lane & 15;
            auto a_tile = [&](const double* T, int ks) { return T[n * 17 + 4 * ks + g]; };         // A[m = n][k]
            auto b_tile = [&](const double* T, int ks) { return T[(4 * ks + g) * 17 + n]; };       // B[k][n]
            auto a_x2 = [&](int ks) {                                                              // X2 restricted to R2 x R2
                const double v = sT1[n * 17 + 4 * ks + g];
                return (n < R2 && 4 * ks + g < R2) ? v : 0.0;
            };
            const d4 zero = d4{0.0, 0.0, 0.0, 0.0};
            if (wave == 0) {
                d4 t10 = zero, x10 = zero, t20 = zero, x20 = zero;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) t10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_tile(L10, ks), b_tile(X0, ks), t10, 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) t20 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_tile(L20, ks), b_tile(X0, ks), t20, 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) x10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_tile(sT0, ks), t10[ks], x10, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) { x10[r] = -x10[r]; xd(1, 0)[(g + 4 * r) * NVS + n] = x10[r]; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) t20 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_tile(L21, ks), x10[ks], t20, 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) x20 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_x2(ks), t20[ks], x20, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (g + 4 * r < R2) xd(2, 0)[(g + 4 * r) * NVS + n] = -x20[r];
            } else if (wave == 1) {
                d4 t21 = zero, x21 = zero;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) t21 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_tile(L21, ks), b_tile(sT0, ks), t21, 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) x21 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_x2(ks), t21[ks], x21, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (g + 4 * r < R2) xd(2, 1)[(g + 4 * r) * NVS + n] = -x21[r];
            } else if (wave == 2) {
                // the three diagonal blocks into the dense X (lower triangles; rows of the last block beyond R2 stay zero)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = g + 4 * r;
                    const double x0 = X0[i * 17 + n], x1 = sT0[i * 17 + n], x2 = sT1[i * 17 + n];
                    if (n <= i) {
                        xd(0, 0)[i * NVS + n] = x0;
                        xd(1, 1)[i * NVS + n] = x1;
                        if (i < R2) xd(2, 2)[i * NVS + n] = x2;
                    }
                }
            } else {
                if (lane < D::NV) sSvec[lane] = (sSv[4 * lane] + sSv[4 * lane + 1]) + (sSv[4 * lane + 2] + sSv[4 * lane + 3]);
                double gm = 0.0;
                for (int c = 0; c < D::NV; ++c) gm = fmax(gm, fabs(sSvec[c]));  // uniform addresses: LDS broadcasts
                if (lane == 0) sSvec[0] = gm;   // every lane of this wavefront has read sSvec[0] (in-order LDS)
            }
        }
        __syncthreads();
        {   // round 6: ALL of P = X_N^T X_N on the matrix cores (six lower 16 x 16 tiles, twelve k-steps each, over the four
            // wavefronts): ~2 k cycles once, where a column formed on demand inside the iteration costs one wavefront ~1.3 k
            // and an instance needs eight to twelve of them.  sP[b * NVS + i] = P[i][b], both triangles.
            static_assert(D::NV <= 48, "three tile rows");
            const int n = hold ? D::NV - 4 : D::NV;
            const int g = lane >> 4, m = lane & 15;
#pragma unroll 1
            for (int t = wave; t < 6; t += D::NWAVES) {
                const int ta = t < 1 ? 0 : (t < 3 ? 1 : 2), tb = t - ta * (ta + 1) / 2;
                const int ca = 16 * ta + m, cb = 16 * tb + m;
                double xa_[12], xb_[12];
#pragma unroll
                for (int ks = 0; ks < 12; ++ks) {
                    const int j = 4 * ks + g;
                    const double va = sXd[j * NVS + ca], vb = sXd[j * NVS + cb];   // (in range of the LDS for every lane)
                    xa_[ks] = (j < n && ca < D::NV) ? va : 0.0;
                    xb_[ks] = (j < n && cb < D::NV) ? vb : 0.0;
                }
                d4 c0 = d4{0.0, 0.0, 0.0, 0.0}, c1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 12; ks += 2) {
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa_[ks], xb_[ks], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa_[ks + 1], xb_[ks + 1], c1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * ta + g + 4 * r;
                    const double pv = c0[r] + c1[r];
                    if (i < D::NV && cb < D::NV) {
                        sSv[cb * NVS + i] = pv;
                        sSv[i * NVS + cb] = pv;
                    }
                }
            }
        }
        __syncthreads();
        if (wave == 0) {
            const XDense<D> xa{sXd};
            dual_active_set<D>(xa, hold, lane, sSv, sQP, sSvec, sVprev, sCfg, sZ, sFlags, ~0ull);
        }
       }
      } else {
        // Schur complement S = L22 L22^T, s = L22 (L^-1 g)_v
        for (int e = tid; e < D::NV * D::NV; e += D::BLOCK) {
            const int r = e / D::NV, c = e % D::NV;
            const int kmax = r < c ? r : c;
            double sum = 0.0;
            for (int k = 0; k <= kmax; ++k)
                sum += Lb[lower_at<D>(D::NU + r, D::NU + k)] * Lb[lower_at<D>(D::NU + c, D::NU + k)];
            sSv[r * (D::NV + 1) + c] = sum;
        }
        if (tid < D::NV) {
            double sum = 0.0;
            for (int k = 0; k <= tid; ++k)
                sum += Lb[lower_at<D>(D::NU + tid, D::NU + k)] * Lb[lower_at<D>(D::NZ, D::NU + k)];
            sSvec[tid] = sum;
        }
        __syncthreads();

        if (wave == 0) {
            const int r = lane < D::NV ? lane : D::NV - 1;  // lanes >= NV shadow the last row (results unused)
            const bool valid = lane < D::NV;
            double row[D::NV];
#pragma unroll
            for (int c = 0; c < D::NV; ++c) row[c] = sSv[r * (D::NV + 1) + c];
            const double svr = sSvec[r];
            const bool fixed = valid && hold && (r >= D::NV - 4);  // v0 is the trailing block
            const double lo = fixed ? sVprev[r & 3] : sCfg[CFG_VMIN];    // constraintsVSMPC.cpp:351-364
            const double hi = fixed ? sVprev[r & 3] : sCfg[CFG_VMAX];
            int state = fixed ? -1 : 0;  // 0 free, -1 at lower, +1 at upper
            double gmax = fabs(svr);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) gmax = fmax(gmax, __shfl_xor(gmax, o));
            const double gtol = 1e-10 * (1.0 + gmax);
            // Iteration 1 of the block-pivoting scheme is the solve with only the hold pin enforced: that is the
            // backward sweep that just ran (its throttles are in sZ).  Apply its flips here instead of repeating
            // the solve; nothing is at a bound yet, so only primal violations can occur.
            double v = sZ[D::NU + r];
            int best, patience = AS_PATIENCE, status = VSMPC_STATUS_MAX_ITER, iters = 1;
            {
                const double tolv = 1e-12 * (1.0 + fabs(v));
                const bool vlo = valid && state == 0 && (v < lo - tolv);
                const bool vhi = valid && state == 0 && (v > hi + tolv);
                best = __popcll(__ballot(vlo || vhi));
                if (vlo || vhi) state = vlo ? -1 : 1;
            }
            for (int it = 1; it < AS_MAX_ITER; ++it) {
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
                int bad = 0;
                const int nfree = __popcll(Fmask);
                if (nfree >= 1 && nfree <= SMALL_SOLVE_MAX) {
                    // deep saturation leaves few free throttles: S_FF v_F = b_F redundantly in registers on
                    // wave-uniform values (S is symmetric: sSv[c * (NV+1) + i] = S[i][c], the layout the solver reads)
                    const double vf = small_spd_solve_n<D::NV + 1>(nfree, sSv, Fmask, b, lane, bad);
                    v = isF ? vf : vb;
                } else {
                // Gaussian elimination without pivoting (SPD), pivot rows broadcast with v_readlane.  Rows of bound
                // throttles are identity rows whose column is zero elsewhere: their pivots are no-ops and are skipped
                // (wave-uniform branch), so the cost follows the number of free throttles
#pragma unroll
                for (int j = 0; j < D::NV; ++j) {
                    if ((Fmask >> j) & 1ull) {
                        const double piv = readlane_f64(a[j], j);
                        bad |= !(piv > 0.0);
                        const double f = (lane > j) ? a[j] * fast_rcp(piv) : 0.0;
                        const double bj = readlane_f64(b, j);
                        b -= f * bj;
#pragma unroll
                        for (int c = j + 1; c < D::NV; ++c) {
                            const double pc = readlane_f64(a[c], j);
                            a[c] -= f * pc;
                        }
                    }
                }
                v = vb;  // bound throttles; free ones follow from the back-substitution
#pragma unroll
                for (int j = D::NV - 1; j >= 0; --j) {
                    if ((Fmask >> j) & 1ull) {
                        const double xj = readlane_f64(b, j) * fast_rcp(readlane_f64(a[j], j));
                        if (lane == j) v = xj;
                        if (lane < j) b -= a[j] * xj;
                    }
                }
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
                if (ninf < best) { best = ninf; patience = AS_PATIENCE; }
                else if (patience > 0) { --patience; }
                else { pick = inf && (lane == 63 - __clzll(imask)); }  // least-index fallback (largest index)
                if (pick) state = vlo ? -1 : (vhi ? 1 : 0);
            }
            if (valid) {
                v = state < 0 ? lo : (state > 0 ? hi : v);  // bound variables sit exactly on their bound
                sZ[D::NU + lane] = v;
            }
            if (lane == 0) { sFlags[1] = status; sFlags[2] = iters; }
        }
      }
}

// ------------------------------------------------------------------------------------------------
// P1s: structured condensing (Dims::STRUCT_P1; executable model: tests/condense_model.py).
//
// The model (systemDynamicsVSMPC.cpp:79-103,288-319,384-429) is a cascade throttles -> jets -> momenta -> CoM / RPY ->
// error integrators whose linear half (p, h_lin, e_pos) and angular half (rpy, h_ang, e_rpy) do not talk to each other,
// and every input reaches a half only as a 3-vector forcing of its momentum rows:  phi_i = Lambda U_{jb(i)} + A_mom T_i.
// Per half, with xi = (x, h, e), Abar_m = I + dt_m K, the condensed Hessian (what constraintsVSMPC.cpp:76-131 and
// costsVSMPC.cpp:166-200 imply once the states are eliminated) is
//     C[r, c] = sum_half sum_i pi_r(i)^T W_c(i),      W_c(i) = dt_i E_h^T nu_c(i + 1),
//     nu_c(m) = Q xi^c_m + Abar_m^T nu_c(m + 1)       (adjoint of column c's own forward trajectory xi^c),
// pi_c(i) the column's forcing profile: Lambda[:, q] while jb(i) = b for the joint column (b, q), A_mom[:, q] tau_i for a
// throttle column (tau = its jet's thrust trajectory from P1a).  A joint block only ever enters through the three momentum
// directions, so 3 generator columns per block (unit forcing e_d) stand for its 8 joint columns: one lane per
// (generator | throttle column | affine column) and half runs the forward recursion with the momentum part of the
// trajectory held in REGISTERS (3 N doubles; x and e are rolled back in the adjoint pass), then the adjoint recursion
// backwards, and leaves in LDS
//     sH [half][pair(bc <= br)][a][d]  = sum_{i in br} W_gen(bc, d)(i)[a]          (KIND 0, generator lanes)
//     sRb[half][c][b][a]               = sum_{i in b} W_c(i)[a]                     (KIND 1, throttle / affine lanes)
//     sW3[half][c][i - 1][a]           = W_c(i)[a],  i >= 1  (tau_0 = 0; short horizons: p1s_contract sums
//                                        sAc[c][i - 1][q] = sum_half A_mom[:, q]^T W_c(i) from it; long horizons add
//                                        their half of sAc straight from the chain with LDS atomics and have no sW3)
// from which p1s_entries forms every entry of C directly in the accumulator layout of the owning wavefront:
//     joint x joint        Lambda[:, qr]^T sH Lambda[:, qc]                               (summed over the halves)
//     throttle x joint     Lambda[:, qc]^T sRb[cr][bc]
//     throttle x throttle  sum_i tau^cc_i sAc[cr][i][q_cc]      (row = affine column: the condensed gradient)
// O(N^2) small 3x3 work (~0.3 MFLOP at the paper horizon) instead of the SYRK over the 18 N weighted sensitivity rows
// (3.8 MFLOP executed).  The affine column carries x0, c and the reference: W_aff(i) = gamma_i.
// ------------------------------------------------------------------------------------------------
// a wave-uniform double moved into scalar registers (v_fma_f64 takes one scalar operand pair): the coefficient matrices of
// a chain cost no vector registers, which is what lets the trajectory stay in them
VS_DEV double uniform_f64(double x) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}

template <class D, int KIND>
VS_DEV void p1s_chain(const DevCfg& cfg, int half, int lane, double* __restrict__ sm) {
    using S = Smem<D>;
    constexpr int N = D::N, HC = D::HC, NV = D::NV;
    const double* sA = sm + S::oA;
    const double* sCfg = sm + S::oCfg;
    double* sH = sm + S::oSH;
    double* sRb = sm + S::oSRb;
    double* sW3 = sm + S::oSW3;
    const int xr0 = half ? 6 : 0, hr0 = half ? 9 : 3, er0 = half ? 23 : 20;   // state rows of this half
    const int yx0 = half ? 6 : 0, yh0 = half ? 9 : 3, ye0 = half ? 15 : 12;   // weighted-row slots (CFG_SQ, reference rows)
    // wave-uniform coefficients, in scalar registers.  A[h, h] = -S(omega) (systemDynamicsVSMPC.cpp:90-91,301-302) is
    // skew-symmetric with a zero diagonal: three numbers, and its transpose is its negative
    double M1[9], qx[3], qh[3], qe[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) M1[3 * r + c] = uniform_f64(sA[(xr0 + r) * NX + hr0 + c]);
        const double sx = sCfg[CFG_SQ + yx0 + r], sh = sCfg[CFG_SQ + yh0 + r], se = sCfg[CFG_SQ + ye0 + r];
        qx[r] = uniform_f64(sx * sx);
        qh[r] = uniform_f64(sh * sh);
        qe[r] = uniform_f64(se * se);
    }
    const double s01 = uniform_f64(sA[(hr0 + 0) * NX + hr0 + 1]), s02 = uniform_f64(sA[(hr0 + 0) * NX + hr0 + 2]),
                 s12 = uniform_f64(sA[(hr0 + 1) * NX + hr0 + 2]);
    // lane -> column.  Long horizons (Dims::NGX > 0): the generator columns 64 .. 3 HC - 1 ride in the lanes behind the
    // affine column of the KIND 1 wavefront of their half (unit forcing while their block is active: the activity takes
    // the place of the thrust trajectory, everything else reads the zeros)
    constexpr int NGX = D::NGX;
    constexpr bool LONG = D::STRUCT_LONG;
    constexpr int NLIVE = KIND == 0 ? (3 * HC < 64 ? 3 * HC : 64) : NV + 1 + NGX;
    const bool live = lane < NLIVE;
    const int col = live ? lane : NLIVE - 1;        // idle lanes shadow the last column and store nothing
    const bool isgen = KIND == 0 || (NGX > 0 && col > NV);
    const int gi = KIND == 0 ? col : (isgen ? 64 + col - (NV + 1) : 0);
    const int gb = gi / 3, gd = gi - 3 * gb;        // generator: joint block, momentum direction
    const bool affl = KIND == 1 && col == NV;       // KIND 1: the affine column
    // x carries x + c_e throughout (e' = x + c_e; the offset is folded into the reference the affine column reads)
    double dir[3];
    double x[3], h[3], e[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if constexpr (KIND == 0) {
            dir[r] = gd == r ? 1.0 : 0.0;
            x[r] = 0.0; h[r] = 0.0; e[r] = 0.0;
        } else {
            const double a_q = sA[(hr0 + r) * NX + 12 + (col & 3)];
            dir[r] = affl ? 0.0 : (isgen ? (gd == r ? 1.0 : 0.0) : a_q);
            const double x0 = sm[S::oIn + VSMPC_IN_X0 + xr0 + r], h0 = sm[S::oIn + VSMPC_IN_X0 + hr0 + r],
                         e0 = sm[S::oIn + VSMPC_IN_X0 + er0 + r];
            const double ce0 = sm[S::oC + er0 + r];
            x[r] = affl ? x0 + ce0 : 0.0;
            h[r] = affl ? h0 : 0.0;
            e[r] = affl ? e0 : 0.0;
        }
    }
    // per-lane operand rows (KIND 1), as offsets into the workgroup's LDS: the jet's thrust trajectory; the affine column
    // reads its forcing A_mom Tbar_k + c_h and the reference where every other column reads zeros (no select in the chain)
    const int tauOff = (KIND == 1 && !affl && !isgen) ? S::oJetT + col * N : S::oSZero;
    const int gaOff = affl ? S::oGA + half * 3 * N : S::oSZero;
    const int refOff = affl ? S::oSRefC : S::oSZero;
    const double* tauRow = sm + tauOff;
    const double* gaRow = sm + gaOff;
    const double* refRow = sm + refOff;
    // long horizons: this half's thrust map and the row of sAc this lane adds to (see Smem::ac_off; the
    // stored stage i' lies at (i' - ac_first) * 4 behind it, so the offset of stage 0 is folded in)
    double Am[3][NTH];
    int acFirst = 0, acOff = 0;
    if constexpr (KIND == 1 && LONG) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int q = 0; q < NTH; ++q) Am[a][q] = sA[(hr0 + a) * NX + 12 + q];   // vector registers: the scalar file is full
        const int cr = col <= NV ? col : NV;
        acFirst = S::ac_first(cr);
        acOff = S::ac_off(cr) - 4 * acFirst;
    }
    // ---- forward: xi_{k+1} = xi_k + dt_k (K xi_k + forcing).  Only the momentum part of the trajectory is kept (3 N
    // doubles; the whole trajectory would be 18 N registers): x and e are rolled BACK in the adjoint pass, which explicit
    // Euler allows exactly up to rounding (x_m = x_{m+1} - dt_m M1 h_m, e_m = e_{m+1} - dt_m (x_m + c_e)).
    // Operands of stage k + 1 are requested at the top of stage k behind an offset the compiler cannot see through:
    // otherwise the loads of ALL stages are hoisted to the top of the unrolled chain and the trajectory gets spilled.
    double hk[N][3];
    double tk_n = 0.0, ga_n[3] = {0.0, 0.0, 0.0};
    if constexpr (KIND == 1) {
        tk_n = tauRow[0];
#pragma unroll
        for (int r = 0; r < 3; ++r) ga_n[r] = gaRow[r];
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double dt = cfg.dt[k];   // kernel argument: a scalar load
        double f[3];
        if constexpr (KIND == 0) {
            const double act = joint_block_of_stage<D>(k) == gb ? 1.0 : 0.0;
#pragma unroll
            for (int r = 0; r < 3; ++r) f[r] = act * dir[r];
        } else {
            double tk = tk_n;
            if constexpr (NGX > 0) tk = isgen ? (joint_block_of_stage<D>(k) == gb ? 1.0 : 0.0) : tk;
#pragma unroll
            for (int r = 0; r < 3; ++r) f[r] = fma(tk, dir[r], ga_n[r]);
            if (k + 1 < N) {
                int zo = 0;
                asm volatile("" : "+v"(zo));
                tk_n = tauRow[k + 1 + zo];
#pragma unroll
                for (int r = 0; r < 3; ++r) ga_n[r] = gaRow[3 * (k + 1) + r + zo];
            }
        }
        double dx[3], dh[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            dx[r] = fma(M1[3 * r + 2], h[2], fma(M1[3 * r + 1], h[1], M1[3 * r] * h[0]));
        }
        dh[0] = fma(s02, h[2], fma(s01, h[1], f[0]));
        dh[1] = fma(s12, h[2], fma(-s01, h[0], f[1]));
        dh[2] = fma(-s12, h[1], fma(-s02, h[0], f[2]));
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            e[r] = fma(dt, x[r], e[r]);           // explicit Euler: the old x (+ c_e)
            x[r] = fma(dt, dx[r], x[r]);
            h[r] = fma(dt, dh[r], h[r]);
        }
        // The state passes through an (empty) volatile statement at every stage boundary: volatile statements keep their
        // order, so stage k + 1 cannot start before stage k is complete.  Without it the instruction selector emits the
        // h chain of all stages first and the x and e chains afterwards, with every intermediate x_k alive in between.
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(e[0]), "+v"(e[1]), "+v"(e[2]));
#pragma unroll
        for (int r = 0; r < 3; ++r) hk[k][r] = h[r];   // h of node k + 1
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- backward: nu(m) = Q w_m + Abar_m^T nu(m + 1), m = N .. 1, w_m = xi_m (minus the reference on the affine column);
    // W(i) = dt_i nu(i + 1)[h].  (x, e) hold node m at the top of step m.
    double nx[3] = {0.0, 0.0, 0.0}, nh[3] = {0.0, 0.0, 0.0}, ne[3] = {0.0, 0.0, 0.0}, bs[3] = {0.0, 0.0, 0.0};
    double rf_n[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if constexpr (KIND == 1) {
        constexpr int rcN = N - 1 < D::NS ? 0 : N - 1 - D::NS;   // reference column of node N (costsVSMPC.cpp:191-200)
#pragma unroll
        for (int r = 0; r < 3; ++r) { rf_n[r] = refRow[rcN * 12 + yx0 + r]; rf_n[3 + r] = refRow[rcN * 12 + yh0 + r]; }
    }
#pragma unroll
    for (int m = N; m >= 1; --m) {
        const int i = m - 1;
        double rf[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) rf[r] = rf_n[r];
        if constexpr (KIND == 1) {
            // reference column of node m - 1; the window only moves at the slow rate, so the fast nodes share column 0
            // and need no reload (costsVSMPC.cpp:191-200)
            const int rc = m - 2 < D::NS ? 0 : m - 2 - D::NS;
            const int rc_cur = m - 1 < D::NS ? 0 : m - 1 - D::NS;
            if (m >= 2 && rc != rc_cur) {
                int zo = 0;
                asm volatile("" : "+v"(zo));
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    rf_n[r] = refRow[rc * 12 + yx0 + r + zo];
                    rf_n[3 + r] = refRow[rc * 12 + yh0 + r + zo];
                }
            }
        }
        if (m < N) {
            const double dtm = cfg.dt[m];
            double t[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) t[r] = fma(M1[6 + r], nx[2], fma(M1[3 + r], nx[1], M1[r] * nx[0]));
            t[0] = fma(-s02, nh[2], fma(-s01, nh[1], t[0]));   // + Sk^T nu_h = - Sk nu_h
            t[1] = fma(-s12, nh[2], fma(s01, nh[0], t[1]));
            t[2] = fma(s12, nh[1], fma(s02, nh[0], t[2]));
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                nx[r] = fma(dtm, ne[r], nx[r]);   // K^T: the x rows of the adjoint collect the e rows (A[e, x] = I)
                nh[r] = fma(dtm, t[r], nh[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            nx[r] = fma(qx[r], KIND == 1 ? x[r] - rf[r] : x[r], nx[r]);
            nh[r] = fma(qh[r], KIND == 1 ? hk[i][r] - rf[3 + r] : hk[i][r], nh[r]);
            ne[r] = fma(qe[r], e[r], ne[r]);
        }
        const double dti = cfg.dt[i];
        if (m >= 2) {   // roll (x, e) back to node m - 1 with h of node m - 1
            // (opaque copies: otherwise the compiler recognises M1 h of the forward pass and keeps all N of them alive
            // -- in scratch -- instead of recomputing, which is the whole point of keeping only h)
            double hp[3] = {hk[i - 1][0], hk[i - 1][1], hk[i - 1][2]};
            asm volatile("" : "+v"(hp[0]), "+v"(hp[1]), "+v"(hp[2]));
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double dxr = fma(M1[3 * r + 2], hp[2], fma(M1[3 * r + 1], hp[1], M1[3 * r] * hp[0]));
                x[r] = fma(-dti, dxr, x[r]);
                e[r] = fma(-dti, x[r], e[r]);
            }
        }
        double w[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { w[a] = dti * nh[a]; bs[a] += w[a]; }
        if constexpr (KIND == 1 && !LONG) {
            if (i >= 1 && live) {
                double* Wp = sW3 + ((half * (NV + 1) + col) * (N - 1) + (i - 1)) * 3;
#pragma unroll
                for (int a = 0; a < 3; ++a) Wp[a] = w[a];
            }
        }
        if constexpr (KIND == 1 && LONG) {
            // A_mom[:, q]^T W_c(i), added to what the other half leaves in the same word (two addends on a zeroed word:
            // the sum does not depend on which arrives first)
            if (i >= 1) {
                if (live && !isgen && i - 1 >= acFirst) {
                    typedef __attribute__((address_space(3))) double lds_double;
                    double* Ap = sm + S::oSAc + acOff + (i - 1) * 4;
#pragma unroll
                    for (int q = 0; q < NTH; ++q) {
                        const double v = fma(Am[2][q], w[2], fma(Am[1][q], w[1], Am[0][q] * w[0]));
                        __builtin_amdgcn_ds_atomic_fadd_f64((lds_double*)(Ap + q), v, 0, 0, false);
                    }
                }
            }
        }
        if (i < HC) {   // i is the first stage of joint block jb(i) = i (the last block spans stages HC-1 .. N-1)
            if (isgen) {
                if (live && i >= gb) {
                    double* Hp = sH + (half * D::NJPAIR + i * (i + 1) / 2 + gb) * 9 + gd;
#pragma unroll
                    for (int a = 0; a < 3; ++a) Hp[3 * a] = bs[a];
                }
            } else {
                if (live) {
                    double* Rp = sRb + ((half * (NV + 1) + col) * HC + i) * 3;
#pragma unroll
                    for (int a = 0; a < 3; ++a) Rp[a] = bs[a];
                }
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) bs[a] = 0.0;
        }
        asm volatile("" : "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nh[0]), "+v"(nh[1]), "+v"(nh[2]), "+v"(ne[0]), "+v"(ne[1]), "+v"(ne[2]));
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(bs[0]), "+v"(bs[1]), "+v"(bs[2]));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// sAc[c][i - 1][q] = sum over the halves of A_mom,half[:, q]^T W_c(i): the throttle x throttle tiles read it as a matrix-core
// operand.  All wavefronts, between the chains and the entries: one (c, i) pair and its four q per thread and round.
template <class D>
VS_DEV void p1s_contract(double* __restrict__ sm, int tid) {
    using S = Smem<D>;
    constexpr int NI = (D::NV + 1) * (D::N - 1);
    constexpr int ROUNDS = (NI + D::BLOCK - 1) / D::BLOCK;
    const double* sA = sm + S::oA;
    const double* sW3 = sm + S::oSW3;
    double* sAc = sm + S::oSAc;
    double w[ROUNDS][6];
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int ci = tid + rd * D::BLOCK, cic = ci < NI ? ci : NI - 1;
#pragma unroll
        for (int a = 0; a < 3; ++a) { w[rd][a] = sW3[cic * 3 + a]; w[rd][3 + a] = sW3[(NI + cic) * 3 + a]; }
    }
    double Am[6][NTH];   // uniform addresses: LDS broadcasts
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < NTH; ++q) { Am[a][q] = sA[(3 + a) * NX + 12 + q]; Am[3 + a][q] = sA[(9 + a) * NX + 12 + q]; }
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int ci = tid + rd * D::BLOCK;
        double v[NTH];
#pragma unroll
        for (int q = 0; q < NTH; ++q) {
            v[q] = Am[0][q] * w[rd][0];
#pragma unroll
            for (int a = 1; a < 6; ++a) v[q] = fma(Am[a][q], w[rd][a], v[q]);
        }
        if (ci < NI) {
#pragma unroll
            for (int q = 0; q < NTH; ++q) sAc[ci * 4 + q] = v[q];
        }
    }
}

// Entries of C = sum_k Y_k^T Y_k for the accumulator tiles wavefront W owns, formed ON THE MATRIX CORES from the small LDS
// arrays the chains leave behind, so that they arrive in the accumulator layout (lane (g, j) holds rows g + 4 r, column j)
// with a handful of LDS reads per tile.  With L = R^T (6 x 6, rows = (half, a), the input matrix of the reduced joint
// unknowns; p0_joint_reduction) a tile row t holds the joint rows 16 t .. 16 t + 15 = unknown (16 t + j) % 6 of block
// (16 t + j) / 6: at most FOUR blocks, the first one blk0(t) = 16 t / 6.  The k index of a product is split as
// k = 4 ks + g  ->  (a6, kb) = (ks, g)   (a6 = (half, a), kb = block within the tile row):
//   joint x joint        D = A B,  A[m][k] = [blk(ti, m) == blk0(ti) + kb] L[a6][unknown(ti, m)],
//                        B[k][n] = (H^(blk0(ti) + kb, blk(tj, n)) L)[a6][unknown(tj, n)]                  6 k-steps
//   throttle x joint     A[m][k] = sRb[half(a6)][row m][blk0(tj) + kb][a],
//                        B[k][n] = [blk(tj, n) == blk0(tj) + kb] L[a6][unknown(tj, n)]                    6 k-steps
//   throttle x throttle  k = (i', q): A[m][k] = sAc[row m][i'][q],  B[k][n] = [q == q_n] tau^n_{i' + 1}    N - 1 k-steps
// Rows / columns of the dummy unknowns (16 t + j >= NUY) get zero operands.  Operands first, matrix instructions
// afterwards: with the loads of a tile right in front of its instructions every tile pays LDS round trips; the
// accumulators are not live yet, so there are registers for the raw operands of a group of tiles at once (the loads are
// pinned in front of the arithmetic).
template <class D, int TPW, int W, bool PIPE = false>
VS_DEV void p1s_entries(d4 (&acc)[TPW], const double* __restrict__ sm, int lane) {
    using S = Smem<D>;
    constexpr int PVT = D::PVT, N = D::N, HC = D::HC, NV = D::NV;
    static_assert(D::NU % 16 == 0, "joint rows are tile aligned");
    const double* sBj = sm + S::oBj;
    const double* sH = sm + S::oSH;
    const double* sRb = sm + S::oSRb;
    const double* sAc = sm + S::oSAc;
    const double* sJetT = sm + S::oJetT;
    // (opaque: the four wave-specialised copies of this function start with the same table loads, which the compiler would
    // otherwise hoist in front of the wave dispatch and, at the 2x horizon, spill across it -- 100 registers of scratch)
    lane = fresh_lane();
    const int j = lane & 15, g = lane >> 4;
    // per lane and tile-row pattern (16 t mod 6 = 0, 4, 2 for t mod 3 = 0, 1, 2): the unknown and the block within the tile
    // row of row / column j, and the six entries L[a6][unknown]
    // LqM: the same entries masked to the k block of this lane ([blk(t, j) == blk0(t) + g]: the A operand of a joint x joint
    // tile, the B operand of a throttle x joint tile), formed once instead of with six selects per tile
    double Lq[3][NJC], LqM[3][NJC];
    int bin[3];
#pragma unroll
    for (int pat = 0; pat < 3; ++pat) {
        const int o = ((16 * pat) % NJC) + j;
        bin[pat] = o / NJC;
        const int un = o - NJC * bin[pat];
#pragma unroll
        for (int a6 = 0; a6 < NJC; ++a6) {
            Lq[pat][a6] = sBj[((a6 < 3 ? 3 : 6) + a6) * NJ + un];
            LqM[pat][a6] = bin[pat] == g ? Lq[pat][a6] : 0.0;
        }
    }
    // tiles in groups of G (the raw operands of a group are all requested before its arithmetic starts)
    // 18 G + 12 G operand registers (doubles) beside the 18 of Lq; long horizons keep finished tiles in registers meanwhile
    constexpr int G = 3;
    constexpr int NGRP = (TPW + G - 1) / G;
    static_for<0, NGRP>([&](auto gcst) __attribute__((always_inline)) {
    constexpr int q0 = decltype(gcst)::value * G;
    constexpr int q1 = q0 + G < TPW ? q0 + G : TPW;
    double raw[G][NJC][3];
    static_for<q0, q1>([&](auto qcst) __attribute__((always_inline)) {
        constexpr TileTab<D, PIPE> tab{};
        constexpr int q = decltype(qcst)::value;
        constexpr int t = q * D::NWAVES + W;
        if constexpr (tab.forms(t, W)) {
            constexpr int ti = tab.ti[t], tj = tab.tj[t];
            if constexpr (ti < PVT) {
                // H^(br, bc), br = blk0(ti) + g (this lane's k block), bc = the block of column j of tile column tj; stored
                // for br >= bc, transposed otherwise (diagonal tiles only).  Blocks beyond the horizon belong to dummy rows /
                // columns whose other operand is zero: clamped into the array.
                constexpr int b0r = (16 * ti) / NJC, b0c = (16 * tj) / NJC;
                const int brr = b0r + g, bcc = b0c + bin[tj % 3];
                const int br = brr < HC ? brr : HC - 1, bc = bcc < HC ? bcc : HC - 1;
                const bool sw = ti == tj && br < bc;
                const int hi = sw ? bc : br, lo = sw ? br : bc;
                const int st = sw ? 3 : 1, sa = sw ? 1 : 3;
                const double* Hp = sH + (hi * (hi + 1) / 2 + lo) * 9;
#pragma unroll
                for (int ks = 0; ks < NJC; ++ks) {
                    const double* Hk = Hp + (ks / 3) * D::NJPAIR * 9 + sa * (ks % 3);
#pragma unroll
                    for (int d = 0; d < 3; ++d) raw[q - q0][ks][d] = Hk[d * st];
                }
            } else if constexpr (tj < PVT) {
                constexpr int b0c = (16 * tj) / NJC;
                const int cr = 16 * (ti - PVT) + j;
                const int crc = cr <= NV ? cr : NV;
                const int bcc = b0c + g, bc = bcc < HC ? bcc : HC - 1;
#pragma unroll
                for (int ks = 0; ks < NJC; ++ks)
                    raw[q - q0][ks][0] = sRb[(((ks / 3) * (NV + 1) + crc) * HC + bc) * 3 + (ks % 3)];
            }
        }
    });
    __builtin_amdgcn_sched_barrier(0);
    double opa[G][NJC], opb[G][NJC];
    static_for<q0, q1>([&](auto qcst) __attribute__((always_inline)) {
        constexpr TileTab<D, PIPE> tab{};
        constexpr int q = decltype(qcst)::value;
        constexpr int t = q * D::NWAVES + W;
#pragma unroll
        for (int ks = 0; ks < NJC; ++ks) { opa[q - q0][ks] = 0.0; opb[q - q0][ks] = 0.0; }
        if constexpr (tab.forms(t, W)) {
            constexpr int ti = tab.ti[t], tj = tab.tj[t];
            // rows / columns of the dummy unknowns exist in the last joint tile row / column only (compile time)
            constexpr bool DUMMY_ROWS = 16 * ti + 16 > D::NUY, DUMMY_COLS = 16 * tj + 16 > D::NUY;
            if constexpr (ti < PVT) {
                const bool okm = 16 * ti + j < D::NUY, okn = 16 * tj + j < D::NUY;
#pragma unroll
                for (int ks = 0; ks < NJC; ++ks) {
                    const int h3 = 3 * (ks / 3);
                    const double hl = fma(raw[q - q0][ks][2], Lq[tj % 3][h3 + 2],
                                          fma(raw[q - q0][ks][1], Lq[tj % 3][h3 + 1], raw[q - q0][ks][0] * Lq[tj % 3][h3]));
                    opa[q - q0][ks] = (!DUMMY_ROWS || okm) ? LqM[ti % 3][ks] : 0.0;   // A: row j of tile row ti sits in k block g
                    opb[q - q0][ks] = (!DUMMY_COLS || okn) ? hl : 0.0;
                }
            } else if constexpr (tj < PVT) {
                const bool okr = 16 * (ti - PVT) + j <= NV;
                const bool okn = 16 * tj + j < D::NUY;
#pragma unroll
                for (int ks = 0; ks < NJC; ++ks) {
                    opa[q - q0][ks] = okr ? raw[q - q0][ks][0] : 0.0;
                    opb[q - q0][ks] = (!DUMMY_COLS || okn) ? LqM[tj % 3][ks] : 0.0;
                }
            }
        }
    });
    __builtin_amdgcn_sched_barrier(0);
    static_for<q0, q1>([&](auto qcst) __attribute__((always_inline)) {
        constexpr TileTab<D, PIPE> tab{};
        constexpr int q = decltype(qcst)::value;
        constexpr int t = q * D::NWAVES + W;
        d4 c = d4{0.0, 0.0, 0.0, 0.0}, c2 = d4{0.0, 0.0, 0.0, 0.0};
        if constexpr (tab.forms(t, W)) {
            constexpr int ti = tab.ti[t], tj = tab.tj[t];
            if constexpr (tj < PVT) {
#pragma unroll
                for (int ks = 0; ks < NJC; ks += 2) {   // two accumulators (a dependent FP64 matrix instruction issues every ~95 cycles)
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(opa[q - q0][ks], opb[q - q0][ks], c, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(opa[q - q0][ks + 1], opb[q - q0][ks + 1], c2, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) c[r] += c2[r];
            }
        }
        acc[q] = c;
    });
    __builtin_amdgcn_sched_barrier(0);
    });
    // throttle x throttle tiles: all operand pairs of a tile are requested before its chain starts.  The k-steps start at
    // the first stage any column of the tile can see (tau_i = 0 up to a column's first stage); a row that does not store
    // an earlier stage (Smem::ac_first) meets only such columns in the lower triangle and reads a zero there.
    static_for<0, TPW>([&](auto qcst) __attribute__((always_inline)) {
        constexpr TileTab<D, PIPE> tab{};
        constexpr int q = decltype(qcst)::value;
        constexpr int t = q * D::NWAVES + W;
        if constexpr (tab.forms(t, W)) {
            constexpr int ti = tab.ti[t], tj = tab.tj[t];
            if constexpr (ti >= PVT && tj >= PVT) {
                constexpr int K0 = tile_first_stage<D>(tj);   // first k-step (stage i' = i - 1)
                constexpr int NK = N - 1 - K0;
                // (lane-derived values are formed afresh here: carried across the joint tiles above they were spilled)
                const int ln = fresh_lane(), j = ln & 15, g = ln >> 4;
                const int cr = 16 * (ti - PVT) + j, cc = 16 * (tj - PVT) + j;
                const bool okr = cr <= NV, okc = cc < NV && g == (cc & 3);
                const int crc = okr ? cr : NV;
                const int rfirst = S::ac_first(crc);
                const double* Ap = sAc + S::ac_off(crc) - 4 * rfirst + g;
                const double* Tp = sJetT + (cc < NV ? cc : 0) * N + 1;
                // operand pairs in chunks of CH k-steps (all of them at short horizons): at the 2x horizon a tile has up to 33
                // k-steps, and 66 operand registers requested at once were spilled to scratch as they arrived
                constexpr int CH = D::STRUCT_LONG ? 8 : NK;
                // two accumulators: a dependent v_mfma_f64_16x16x4_f64 issues every ~95 cycles, independent ones every 64
                // (tools/microbench/lat_probe.hip)
                d4 c = d4{0.0, 0.0, 0.0, 0.0}, c2 = d4{0.0, 0.0, 0.0, 0.0};
                static_for<0, (NK + CH - 1) / CH>([&](auto ccst) __attribute__((always_inline)) {
                    constexpr int k0 = decltype(ccst)::value * CH;
                    constexpr int kn = k0 + CH < NK ? CH : NK - k0;
                    double av[kn], bv[kn];
#pragma unroll
                    for (int ks = 0; ks < kn; ++ks) {
                        const int ip = K0 + k0 + ks;
                        // (a short row does not store the stages before rfirst: the load then hits the row in front of it --
                        // always inside the workgroup's LDS -- and is discarded below; a conditional load would be a branch)
                        av[ks] = Ap[4 * ip];
                        bv[ks] = Tp[ip];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < kn; ++ks) {
                        const int ip = K0 + k0 + ks;
                        const bool stored = !(D::STRUCT_LONG && ip < S::AC_NSH) || ip >= rfirst;
                        av[ks] = (okr && stored) ? av[ks] : 0.0;
                        bv[ks] = okc ? bv[ks] : 0.0;
                    }
#pragma unroll
                    for (int ks = 0; ks < kn; ++ks) {
                        if ((k0 + ks) & 1) c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], bv[ks], c2, 0, 0, 0);
                        else c = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], bv[ks], c, 0, 0, 0);
                    }
                });
#pragma unroll
                for (int r = 0; r < 4; ++r) c[r] += c2[r];
                acc[q] = c;
            }
        }
    });
}

// ------------------------------------------------------------------------------------------------
// the solve kernel
// ------------------------------------------------------------------------------------------------
// Kernel-argument block as it lies in the kernarg segment.  Everything but `in` and `batch` is needed late (outputs) or
// rarely (debug dumps, stamps): those are re-read from the kernarg segment where they are used instead of being held in
// (and spilled from) scalar registers for the whole kernel.  late_args() returns the segment pointer behind an opaque
// barrier, so the compiler cannot hoist the loads to the top of the kernel.
struct SolveArgs {
    DevCfg cfg;
    const double* in;
    int batch;
    int pad_;
    double* xout;
    double* fmout;
    int* status_out;
    int* iters_out;
    double* dbgM;
    double* dbgL;
    unsigned long long* stamps;
};
VS_DEV const SolveArgs* late_args() {
    const SolveArgs* ka = reinterpret_cast<const SolveArgs*>(
        (const void*)__builtin_amdgcn_kernarg_segment_ptr());
    asm volatile("" : "+s"(ka));
    return ka;
}

// STAMPS = true is the diagnostic build: thread 0 records s_memtime at every phase boundary into
// `stamps` (its own buffer, never read by the kernel).  The shipped instantiation has STAMPS = false.
// FORM selects how P1 condenses: 0 = sensitivity recursion + SYRK on the matrix cores (every horizon), 1 = structured
// condensing (P1s, horizons with Dims::STRUCT_P1; the default there).  Everything from P2 on is the same code; the two
// forms agree to rounding (different summation order), which tests/test_gpu_parity.py checks on the device.
template <class D, bool STAMPS, int FORM = 0, bool PLDS = false>
__global__ __launch_bounds__(D::BLOCK, D::WG_PER_CU) void solve_kernel(DevCfg cfg, const double* __restrict__ in, int batch,
                                                         double* xout_, double* fmout_, int* status_out_,
                                                         int* iters_out_, double* dbgM_, double* dbgL_,
                                                         unsigned long long* stamps_) {
    // (the trailing parameters are read through late_args(), see SolveArgs)
#define VS_STAMP(i)                                                                         \
    do {                                                                                    \
        if constexpr (STAMPS) {                                                             \
            unsigned long long* st_ = late_args()->stamps;                                  \
            if (tid == 0 && st_ != nullptr) st_[size_t(blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
        }                                                                                   \
    } while (0)
    unsigned long long t_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long t_mark = 0, stamp_t1 = 0, rt0 = 0;
    if constexpr (STAMPS) rt0 = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz clock: wall time of this instance
#define VS_TIC()                                                     \
    do {                                                             \
        if constexpr (STAMPS) t_mark = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define VS_TOC(i)                                                          \
    do {                                                                   \
        if constexpr (STAMPS) {                                            \
            const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
            t_acc[i] += t_now - t_mark;                                    \
            t_mark = t_now;                                                \
        }                                                                  \
    } while (0)
    using S = Smem<D>;
    constexpr bool FUSED_DISPATCH = FORM == 1;   // entries + P2 + P3 behind one wave dispatch (see P1)
    // the structured form runs P3 pipelined: wavefront 0 factors the panels, wavefronts 1..3 hold all tiles (TileTab, cholesky_wave)
#ifndef VS_P3_PIPE
#define VS_P3_PIPE 1
#endif
    constexpr bool PIPE = FORM == 1 && VS_P3_PIPE && VS_PANEL_DPP;
    // steps of the joint reduction that run in P0 (wavefront 3); the rest follows a generator chain in P1.  Long horizons
    // have ~9 k cycles of slack behind the generator chains, short ones ~3 k.
    constexpr int QR_P0_STEPS = D::STRUCT_LONG ? 1 : 2;
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
    double* sF = smem + S::oF;
    double* sDt = smem + S::oDt;
    double* sCfg = smem + S::oCfg;
    int* sFlags = reinterpret_cast<int*>(smem + S::oFlags);  // [0] numerical failure, [1] status, [2] iters, [3] bound violated
    double* sY = smem + S::oY;
    double* Lb = smem + S::oM;        // tile storage: ring of two panel columns + throttle corner (see Dims)
    double* sXinv = smem + S::oXinv;  // inverses of the joint diagonal tiles and of the first throttle tile
    double* sQP = smem + S::oQP;      // dual box QP work arrays
    double* sU = smem + S::oU;        // register back-substitution: per-wavefront partial sums, scratch

    // Work-item ids are RE-DERIVED at every phase boundary (VS_REFRESH_IDS: lane from v_mbcnt behind an opaque operand,
    // wave from its scalar register) instead of being carried through the kernel: a value that is live from the first
    // to the last instruction is the register allocator's favourite spill candidate, and every reload from scratch is a
    // global-memory round trip on the critical path of a latency-bound workgroup.
    const int inst = blockIdx.x;
    if (inst >= batch) return;
    int tid = threadIdx.x;
    int lane = tid & 63;
#ifdef VS_ROLE_ROT   // measurement builds: rotate the roles of the hardware wavefronts with the workgroup index
    const int wave = __builtin_amdgcn_readfirstlane(((tid >> 6) + ((inst >> 8) & 3)) & 3);
    tid = (wave << 6) | lane;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: wave-dependent branches become s_cbranch
#endif
    // (Rotating which hardware wavefront plays which role with the workgroup index -- so that the serial role-0 phases of two
    // co-resident workgroups do not share a SIMD -- measured no different at batch 4096: 392.3 us against 388.1.)
#define VS_REFRESH_IDS()                                                                     \
    do {                                                                                     \
        unsigned m_ = ~0u;                                                                   \
        asm volatile("" : "+s"(m_));                                                         \
        lane = int(__builtin_amdgcn_mbcnt_hi(m_, __builtin_amdgcn_mbcnt_lo(m_, 0u)));        \
        tid = (wave << 6) | lane;                                                            \
    } while (0)

    // tiles of the lower triangle are dealt round-robin to the wavefronts: tile t -> wave t % NWAVES, slot t / NWAVES
    // (stage-sorted table in constant memory, padded with never-active dummies)
    constexpr int TPW = TileTab<D, PIPE>::TPW;

    VS_STAMP(0);
    // ---------------------------------------------------------------- P0
    {   // 16 B per lane: the record stride (NIN doubles) and the LDS base are multiples of 16 B
        static_assert(D::NIN % 2 == 0 && D::NVAR % 2 == 0 && D::NXS % 2 == 0 && D::NUO % 2 == 0, "double2 I/O");
        // the record's HBM round trip (~1 us) is overlapped with the LDS initialisation: loads first, dependent stores last
        static_assert(D::NIN / 2 <= D::BLOCK, "one 16-byte load per thread covers the record");
        const double2* in2 = reinterpret_cast<const double2*>(in + size_t(inst) * D::NIN);
        double2* sIn2 = reinterpret_cast<double2*>(sIn);
        double2 rec = make_double2(0.0, 0.0);
        if (tid < D::NIN / 2) rec = in2[tid];
        if (tid < 4) sFlags[tid] = 0;
        if (tid < D::N) sDt[tid] = cfg.dt[tid];
        if (tid >= 64 && tid < 64 + NWROWS) sCfg[CFG_SQ + tid - 64] = cfg.sq[tid - 64];
        if (tid >= 96 && tid < 96 + NJ) sCfg[CFG_WJ + tid - 96] = cfg.wj[tid - 96];
        if (tid == 128) {
            sCfg[CFG_WREG] = cfg.w_reg; sCfg[CFG_WTHR] = cfg.w_thr; sCfg[CFG_WINIT] = cfg.w_init;
            sCfg[CFG_VMIN] = cfg.vmin; sCfg[CFG_VMAX] = cfg.vmax;
        }
        for (int i = tid; i < NX * NX + NX * NJ + NX * NTH + 28; i += D::BLOCK) sA[i] = 0.0;  // A,Bj,Bt,c contiguous
        if (tid < D::NIN / 2) sIn2[tid] = rec;
    }
    __syncthreads();
    p0_linearize<D, false, false, false>(cfg.use_jet, sIn, sA, sBj, sBt, sC, sVprev, tid, D::BLOCK);  // barrier: after P1a below
    // joint reduction: 6 unknowns per joint block instead of 8 (NJC).  The SYRK form needs the reduced input matrix at the top
    // of its recursion: all six steps here, in wavefront 3, which has only copies to do in P0 (5.6 k cycles, of which ~2.8 k
    // lengthen P0).  The structured form needs it for the tile entries only: three steps here (hidden), the other three in a
    // generator wavefront after its chain, in the ~3 k cycles it would otherwise wait for the throttle wavefronts (below).
    if (wave == 3) {
        if constexpr (FORM == 1) p0_joint_reduction<D, 0, QR_P0_STEPS>(smem, lane);
        else p0_joint_reduction<D>(smem, lane);
    }

    // P1a: jet sub-system.  The model is a cascade (jets -> momenta -> CoM / RPY -> integrators) and the jets are
    // decoupled from each other, so of the condensed columns only the throttle columns (one jet each) and the affine
    // column (four) carry a non-zero thrust sensitivity: NV + 4 two-state recursions over the whole horizon, one per
    // lane, whose thrust trajectories T_k go to LDS.  The momentum recursion below then needs one scalar per column and
    // stage instead of eight jet states, their input vectors and twelve coefficients in every thread -- which is what
    // brings P1 under the 256 registers a wavefront gets when two workgroups share a CU.   (systemDynamicsVSMPC.cpp:384-429)
    constexpr int NJROW = D::NV + NTH + 1, ZROW = D::NV + NTH;  // + an all-zero row for the joint and padding columns
    double* sJetT = sXinv;                    // [NJROW][N]; the X tiles are not written before P3
    double* sGA = sJetT + NJROW * D::N;       // [2][N][3]: A_mom T_k of the affine column, per half
    static_assert(D::NV + NTH <= 64 && NJROW * D::N + 6 * D::N <= S::NXT * D::TS, "jet trajectories fit the X region");
    // Runs in the wavefront whose lanes 0..3 linearised the jets (p0_linearize), straight behind that, while the other
    // wavefronts finish their pieces of P0: LDS operations of one wavefront execute in order, no barrier needed.
    if (wave == 1) {
        if (lane < D::NV + NTH) {
            const bool affl = lane >= D::NV;
            const int i = affl ? lane - D::NV : (lane & 3);
            const int blk = affl ? -1 : v_block_of_internal<D>(lane);
            const double jon = sA[(12 + i) * NX + 16 + i], ja = sA[(16 + i) * NX + 12 + i], jb = sA[(16 + i) * NX + 16 + i];
            const double c12 = sC[12 + i], c16 = sC[16 + i], b12 = sBt[(12 + i) * NTH + i], b16 = sBt[(16 + i) * NTH + i];
            const double t0 = sIn[VSMPC_IN_X0 + 12 + i], td0 = sIn[VSMPC_IN_X0 + 16 + i];
            const double bT = affl ? c12 : b12, bTd = affl ? c16 : b16;
            double T = affl ? t0 : 0.0, Td = affl ? td0 : 0.0;
#pragma unroll
            for (int k = 0; k < D::N; ++k) {
                sJetT[lane * D::N + k] = T;   // the momentum rate of stage k sees T_k (explicit Euler)
                const double mT = (affl || throttle_block_of_stage<D>(k) == blk) ? 1.0 : 0.0;
                const double dT = fma(jon, Td, mT * bT);
                const double dTd = fma(ja, T, fma(jb, Td, mT * bTd));
                const double dt = sDt[k];
                T = fma(dt, dT, T);
                Td = fma(dt, dTd, Td);
            }
        }
        for (int k = lane; k < D::N; k += 64) sJetT[ZROW * D::N + k] = 0.0;
        if constexpr (FORM == 1) {
            // P1s: the affine column's whole momentum forcing A_mom Tbar_k + c_h, straight behind the trajectories it reads (same
            // wavefront: LDS operations stay in order) and from the record instead of the linearisation the other wavefronts
            // are still writing -- A_mom as p0_linearize copies it, c_h = alpha m R^T g as it forms it -- so that P0 needs no
            // second block behind a second barrier (v30; was: all threads, after the barrier below, then another one)
            const double am = sIn[VSMPC_IN_ALPHA] * sIn[VSMPC_IN_MASS];
            const double* R = sIn + VSMPC_IN_WRB;
            const double* gr = sIn + VSMPC_IN_GRAV;
            for (int e = lane; e < 6 * D::N; e += 64) {
                const int h = e / (3 * D::N), k = (e / 3) % D::N, r = e % 3;
                double g = 0.0;
#pragma unroll
                for (int c = 0; c < NTH; ++c) g = fma(sIn[VSMPC_IN_AMOM + (3 * h + r) * NTH + c], sJetT[(D::NV + c) * D::N + k], g);
                const double ch = h == 0 ? am * (R[0 + r] * gr[0] + R[3 + r] * gr[1] + R[6 + r] * gr[2]) : 0.0;
                sGA[e] = g + ch;
            }
        }
    }
    if constexpr (FORM == 1) {
        // the zeros and the reference window (+ c_e = -p_ref / -rpy_init on the CoM / RPY rows, as p0_linearize writes it) by the
        // wavefront whose lane 0 does the scalar CoM / gravity piece of the linearisation: none of it waits for anything
        if (wave == 2) {
            for (int e = lane; e < S::sizeZero; e += 64) smem[S::oSZero + e] = 0.0;
            for (int e = lane; e < 12 * D::NREF; e += 64) {
                const int row = e % 12;
                const double off = row < 3 ? -sIn[VSMPC_IN_PREF + row] : ((row >= 6 && row < 9) ? -sIn[VSMPC_IN_RPYINIT + row - 6] : 0.0);
                smem[S::oSRefC + e] = sIn[VSMPC_IN_XREF + e] + off;
            }
        }
    }
    if constexpr (FORM == 1 && D::STRUCT_LONG) {   // the chains ADD into sAc (30 KB at the 2x horizon: all threads)
        for (int e = tid; e < S::sizeAc; e += D::BLOCK) smem[S::oSAc + e] = 0.0;
    }
    __syncthreads();   // ends P0 and the jet trajectories
    if constexpr (FORM != 1) {
        for (int e = tid; e < 6 * D::N; e += D::BLOCK) {
            const int h = e / (3 * D::N), k = (e / 3) % D::N, r = e % 3, row = (h ? 9 : 3) + r;
            double g = 0.0;
#pragma unroll
            for (int c = 0; c < NTH; ++c) g = fma(sA[row * NX + 12 + c], sJetT[(D::NV + c) * D::N + k], g);
            sGA[e] = g;
        }
        __syncthreads();
    }

    VS_STAMP(1);
    VS_REFRESH_IDS();
    if constexpr (STAMPS) stamp_t1 = __builtin_amdgcn_s_memtime();
    // ---------------------------------------------------------------- P1 condense
    d4 acc[TPW];
    if constexpr (FORM == 1) {
        // P1s: structured condensing.  Generator lanes on wavefronts 0 / 1 (linear / angular half), throttle and affine
        // columns on wavefronts 2 / 3; no communication inside the chains, one barrier, then the entries of the owned
        // tiles straight into the accumulator registers.
        static_assert(D::STRUCT_P1, "structured condensing is instantiated for horizons with Dims::STRUCT_P1");
        VS_TIC();
        if (wave < 2) {
            p1s_chain<D, 0>(cfg, wave, lane, smem);
            if (wave == 0) p0_joint_reduction<D, QR_P0_STEPS, NJC>(smem, lane);   // (second half; writes Bj and sQR, which nobody touches
                                                                        // before the barrier below)
        } else {
            p1s_chain<D, 1>(cfg, wave - 2, lane, smem);
        }
        VS_TOC(0);
        __syncthreads();
        VS_TOC(1);
        if constexpr (!D::STRUCT_LONG) {
            p1s_contract<D>(smem, tid);
            __syncthreads();
        }
        VS_TOC(3);
        if constexpr (FUSED_DISPATCH) {
            // the shipped structured form: ONE wave dispatch for the entries, P2 and P3 -- the accumulator tiles are born and
            // factored inside one branch, no join in between (at a join the allocator moved the tiles of a wavefront through
            // the vector registers, and at the 2x horizon pushed other values into scratch to make room)
            // (The diagnostic instantiation takes the same path; its dumps -- condensed Hessian before, factor after P3 -- and the
            // stamps of the P1 / P2 / P3 boundaries sit inside the branch.)
            auto tail = [&](auto wcst) __attribute__((always_inline)) {
                constexpr int W = decltype(wcst)::value;
                p1s_entries<D, TPW, W, PIPE>(acc, smem, lane);
                VS_TOC(2);
                __syncthreads();   // the LDS arrays of P1s lie under the ring P3 is about to fill
                VS_STAMP(2);
                double* dbgLw = nullptr;
                if constexpr (STAMPS) {
                    constexpr TileTab<D, PIPE> tab{};
                    double* dbgM = late_args()->dbgM;
                    double* dbgL = late_args()->dbgL;
                    dbgLw = dbgL != nullptr ? dbgL + size_t(inst) * D::NP * D::NP : nullptr;
                    if (dbgM != nullptr) {   // debug/parity only: the augmented condensed Hessian before factorisation, from registers
                        const int ln = fresh_lane();
                        static_for<0, TPW>([&](auto qcst) __attribute__((always_inline)) {
                            constexpr int q = decltype(qcst)::value;
                            constexpr int t = q * D::NWAVES + W;
                            if constexpr (tab.forms(t, W)) {
                                constexpr int ti_q = tab.ti[t], tj_q = tab.tj[t];
                                d4 tmp = acc[q];
                                if constexpr (ti_q == tj_q || ti_q >= D::PVT) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r)
                                        tmp[r] += input_cost_term<D>(sCfg, smem + S::oQR + S::QR_GY, sVprev, 16 * ti_q + (ln >> 4) + 4 * r,
                                                                     16 * tj_q + (ln & 15));
                                }
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int gr = 16 * ti_q + (ln >> 4) + 4 * r, gc = 16 * tj_q + (ln & 15);
                                    if (gc <= gr) dbgM[size_t(inst) * D::NP * D::NP + size_t(gr) * D::NP + gc] = tmp[r];
                                }
                            }
                        });
                    }
                }
                VS_STAMP(3);
                if constexpr (D::WG_PER_CU == 1) pin_tiles_agpr<TPW>(acc);
                const int ln = fresh_lane();
                cholesky_wave<D, TPW, W, STAMPS, PLDS, PIPE>(sCfg, acc, Lb, sInvD, smem + S::oQR + S::QR_GY, sVprev, sFlags, sXinv, sW, dbgLw, ln,
                                                      (ln >> 4) * 17 + (ln & 15), (ln & 15) * 17 + (ln >> 4), sZ);
                if constexpr (STAMPS) {
                    if (dbgLw != nullptr) {  // debug/parity only: the factor; diagonal tiles were written while they were panels
                        constexpr TileTab<D, PIPE> tab{};
                        const int l2 = fresh_lane();
                        static_for<0, TPW>([&](auto qcst) __attribute__((always_inline)) {
                            constexpr int q = decltype(qcst)::value;
                            constexpr int t = q * D::NWAVES + W;
                            if constexpr (tab.holds(t, W)) {
                                constexpr int ti_q = tab.ti[t], tj_q = tab.tj[t];
                                if constexpr (tj_q < D::PVT && ti_q > tj_q) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r)
                                        dbgLw[size_t(16 * ti_q + (l2 >> 4) + 4 * r) * D::NP + 16 * tj_q + (l2 & 15)] = acc[q][r];
                                }
                            }
                        });
                    }
                }
            };
            switch (wave) {
                case 0: tail(std::integral_constant<int, 0>{}); break;
                case 1: tail(std::integral_constant<int, 1>{}); break;
                case 2: tail(std::integral_constant<int, 2>{}); break;
                default: tail(std::integral_constant<int, 3>{}); break;
            }
        } else {
        switch (wave) {
            case 0: p1s_entries<D, TPW, 0, PIPE>(acc, smem, lane); break;
            case 1: p1s_entries<D, TPW, 1, PIPE>(acc, smem, lane); break;
            case 2: p1s_entries<D, TPW, 2, PIPE>(acc, smem, lane); break;
            default: p1s_entries<D, TPW, 3, PIPE>(acc, smem, lane); break;
        }
        VS_TOC(2);
        __syncthreads();   // the LDS arrays of P1s lie under the ring P3 is about to fill
        }
    } else {
#pragma unroll
        for (int q = 0; q < TPW; ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
        // P1b: thread (half, c): half 0 = linear part (p, h_lin, e_pos), half 1 = angular part (rpy, h_ang, e_rpy) of the
        // condensed columns c, c + 128, ... (CPT of them; one at the paper horizon).  Same code, different coefficient rows.
        constexpr int CPT = D::CPT;
        // every wavefront runs the recursion of a pass, then its share of the SYRK
        const int pw = wave;
        const int ptid = tid;
        const int half = pw / (D::NWAVES / 2);  // scalar
        const int xr0 = half ? 6 : 0, hr0 = half ? 9 : 3, er0 = half ? 23 : 20;  // state rows
        const int yx0 = half ? 6 : 0, yh0 = half ? 9 : 3, ye0 = half ? 15 : 12;  // weighted-row slots

        // coefficient rows of this half: wave-uniform LDS broadcasts, re-read at the top of every pass so that they are
        // dead during the matrix-core section (the accumulator tiles stay in registers for the whole of P1..P5)
        double M1[9], Sk[9], Ce[3], sqx[3], sqh[3], sqe[3];
        auto load_coeffs = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    M1[3 * r + c] = sA[(xr0 + r) * NX + hr0 + c];
                    Sk[3 * r + c] = sA[(hr0 + r) * NX + hr0 + c];
                }
                Ce[r] = sC[er0 + r];
                sqx[r] = sCfg[CFG_SQ + yx0 + r];
                sqh[r] = sCfg[CFG_SQ + yh0 + r];
                sqe[r] = sCfg[CFG_SQ + ye0 + r];
            }
        };
        // MFMA operand addresses: lane l reads Y[4 ks + (l >> 4)][16 tile + (l & 15)]; the k-step enters as an
        // immediate offset of ds_read_b64
        const int ylane = (lane >> 4) * D::YS + (lane & 15);
        constexpr int NPASS = (D::N + 1) / 2;
        if constexpr (STAMPS) { t_mark = stamp_t1; VS_TOC(3); }  // P1 set-up

        // The sensitivity recursion over the whole horizon + the SYRK into all accumulator slots of this wavefront.
        // Y is single-buffered (two barriers per pass): the second buffer is what kept a second workgroup off the CU,
        // and a co-resident workgroup fills the recursion's bubbles far better than the look-ahead did.
        int col[CPT], kind[CPT], blk[CPT];   // kind: 0 joint column, 1 throttle column, 2 affine column, 3 pad
        double aff[CPT], xs[CPT][3], hs[CPT][3], es[CPT][3], bha[CPT][3];
        const double* jetT[CPT];
        const double* gaT = sGA + half * 3 * D::N;
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) {
            const int c = cc * D::PCOLS + ptid % D::PCOLS;
            int comp = 0;
            col[cc] = c;
            kind[cc] = 3;
            blk[cc] = 0;
            if (c < D::NUY) { kind[cc] = 0; blk[cc] = c / NJC; comp = c - NJC * blk[cc]; }   // (dummy unknowns: kind 3)
            else if (c >= D::NU && c < D::NZ) { kind[cc] = 1; blk[cc] = v_block_of_internal<D>(c - D::NU); comp = (c - D::NU) & 3; }
            else if (c == D::NZ) { kind[cc] = 2; }
            // unconditional loads (every address is valid for every column), selected afterwards: conditional loads
            // become branches and the LDS latencies add up instead of overlapping
            const int jc = comp & 7, tc = comp & 3;
            const bool aff_col = kind[cc] == 2, jnt_col = kind[cc] == 0, thr_col = kind[cc] == 1;
            aff[cc] = aff_col ? 1.0 : 0.0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double x0 = sIn[VSMPC_IN_X0 + xr0 + r], h0 = sIn[VSMPC_IN_X0 + hr0 + r], e0 = sIn[VSMPC_IN_X0 + er0 + r];
                const double bj = sBj[(hr0 + r) * NJ + jc], ch = sC[hr0 + r], at = sA[(hr0 + r) * NX + 12 + tc];
                xs[cc][r] = aff_col ? x0 : 0.0;
                hs[cc][r] = aff_col ? h0 : 0.0;
                es[cc][r] = aff_col ? e0 : 0.0;
                // one input vector per column: joint column -> Bj column (scaled by the block activity), affine column -> c
                // (always on), throttle column -> thrust map column of the one jet it drives (scaled by that jet's T_k)
                bha[cc][r] = jnt_col ? bj : (aff_col ? ch : (thr_col ? at : 0.0));
            }
            // this column's thrust trajectory: a throttle column's own jet, the zero row otherwise (the affine column's
            // four jets enter through sGA)
            jetT[cc] = sJetT + (thr_col ? c - D::NU : ZROW) * D::N;
        }
        // wave-uniform tile coordinates of the slots, packed two slots per scalar register (see TilePack)
        unsigned tpk[TilePack<D>::NWORDS];
#pragma unroll
        for (int k = 0; k < TilePack<D>::NWORDS; ++k) tpk[k] = kTilePack<D>.w[pw][k];
        const double* ybase = sY + ylane;   // + the buffer of the pass (latency form: two Y buffers)
        constexpr bool UNROLLED = VS_SYRK_UNROLL && TPW <= VS_UNROLL_TPW;
        // In a rolled pass loop the operand addresses are loop invariants: the compiler hoists all 2 TPW of them out of the
        // loop and, at 30 slots, spills them -- ~4k cycles of scratch reloads per pass (measured).  The packed word is
        // therefore made opaque where it is used: one shift-and-add per operand address, in place.
        auto slot_word = [&](int q) __attribute__((always_inline)) {
            unsigned w = tpk[q >> 1];
            if constexpr (!UNROLLED) asm volatile("" : "+s"(w));
            return w;
        };
        auto slot_a = [&](int q) __attribute__((always_inline)) { return ybase + 16 * int((slot_word(q) >> (16 * (q & 1))) & 0xffu); };
        auto slot_b = [&](int q) __attribute__((always_inline)) { return ybase + 16 * int((slot_word(q) >> (16 * (q & 1) + 8)) & 0xffu); };
        // one pass of the recursion (nodes 2m, 2m + 1 -> the Y buffer sYm) and one pass of the SYRK (Y buffer behind ybase)
        auto rec_pass = [&](int m, int nnodes, double* sYm) __attribute__((always_inline)) {
            load_coeffs();
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                if (par >= nnodes) break;  // the last pass of an odd horizon has one node
                const int k = 2 * m + par;  // stage k -> node k+1
                const double dt = sDt[k];
                // reference of this node (affine column only; column map costsVSMPC.cpp:191-200) and the affine column's
                // thrust forcing, requested before the recursion so that the LDS latency is spent under it
                const int rc = k < D::NS ? 0 : k - D::NS;
                const double* xr = sIn + VSMPC_IN_XREF + rc * 12;  // uniform address: LDS broadcast
                double xrx[3], xrh[3], ga[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) { xrx[r] = xr[yx0 + r]; xrh[r] = xr[yh0 + r]; ga[r] = gaT[3 * k + r]; }
#pragma unroll
                for (int cc = 0; cc < CPT; ++cc) {
                    const double tk = jetT[cc][k];
                    const bool actJ = (kind[cc] == 0 && joint_block_of_stage<D>(k) == blk[cc]) || kind[cc] == 2;
                    // input activity as a 0/1 factor inside the multiply-adds (a 64-bit select costs two instructions)
                    const double scale = kind[cc] == 1 ? tk : (actJ ? 1.0 : 0.0);
                    double dx[3], dh[3], de[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        double a0 = M1[3 * r] * hs[cc][0], a1 = scale * bha[cc][r], a2 = aff[cc] * ga[r];
#pragma unroll
                        for (int c = 1; c < 3; ++c) a0 = fma(M1[3 * r + c], hs[cc][c], a0);
#pragma unroll
                        for (int c = 0; c < 3; ++c) a1 = fma(Sk[3 * r + c], hs[cc][c], a1);
                        dx[r] = a0;
                        dh[r] = a1 + a2;
                        de[r] = fma(aff[cc], Ce[r], xs[cc][r]);
                    }
#pragma unroll
                    for (int r = 0; r < 3; ++r) { xs[cc][r] += dt * dx[r]; hs[cc][r] += dt * dh[r]; es[cc][r] += dt * de[r]; }
                    // Y rows of this node: sqrt(Q) (S_k - xref_k on the affine column); column map costsVSMPC.cpp:191-200
                    if (CPT == 1 || col[cc] < D::NP) {
                        double* Yn = sYm + 18 * par * D::YS + col[cc];
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            const double vx = fma(-aff[cc], xrx[r], xs[cc][r]);  // aff = 1 on the affine column, else 0
                            const double vh = fma(-aff[cc], xrh[r], hs[cc][r]);
                            Yn[(yx0 + r) * D::YS] = sqx[r] * vx;
                            Yn[(yh0 + r) * D::YS] = sqh[r] * vh;
                            Yn[(ye0 + r) * D::YS] = sqe[r] * es[cc][r];
                        }
                    }
                }
            }
            if (nnodes == 1)  // rows 18,19 of the last, single-node pass (k-step 4 reads rows 16..19)
                for (int i = ptid; i < 2 * D::YS; i += D::BLOCK) sYm[18 * D::YS + i] = 0.0;
        };
        // Short horizons: the pass loop is UNROLLED and the number of slots a pass runs is a compile-time constant, the same
        // for the four wavefronts (the maximum over them: a wavefront with fewer active tiles multiplies columns of Y
        // that are still exactly zero).  The SYRK of a pass is then straight-line code -- a branch per slot, or any
        // other control flow around the chains, makes the register allocator move whole accumulator tiles at the joins
        // (measured: ~1.1k cycles per pass).  The wavefront with the most active tiles sets the pace either way.
        auto syrk_fixed = [&](auto nks_c, auto nact_c) __attribute__((always_inline)) {
            constexpr int NKS = decltype(nks_c)::value, NACT = decltype(nact_c)::value;
            if constexpr (NACT > 0) {
                double ha[SYRK_DIST], hb[SYRK_DIST];
#pragma unroll
                for (int ks = 0; ks < SYRK_DIST; ++ks) {
                    ha[ks] = slot_a(NACT - 1)[ks * 4 * D::YS];
                    hb[ks] = slot_b(NACT - 1)[ks * 4 * D::YS];
                }
#pragma unroll
                for (int q = NACT - 1; q >= 0; --q) {
                    const int qn = q > 0 ? q - 1 : 0;
                    syrk_slot<D, NKS, VS_SYRK_TIED, true>(acc[q], slot_a(q), slot_b(q), ha, hb, slot_a(qn), slot_b(qn));
#ifdef VS_DIAG_SPLIT   // measurement builds: time of the first chain of every pass -> sub-phase 3
                    if (q == NACT - 1) VS_TOC(3);
#endif
                }
            }
        };
        if constexpr (UNROLLED) {
            static_for<0, NPASS>([&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value;
                constexpr int nnodes = (2 * m + 1 < D::N) ? 2 : 1;
                VS_TIC();
                rec_pass(m, nnodes, sY);
                VS_TOC(0);
                __syncthreads();
                VS_TOC(1);
                syrk_fixed(std::integral_constant<int, nnodes == 2 ? 9 : 5>{}, std::integral_constant<int, nact_max<D>(m)>{});
#ifdef VS_DIAG_PASS
                if (m != VS_DIAG_PASS) { VS_TIC(); } else
#endif
                VS_TOC(2);
                __syncthreads();  // single Y buffer: the next pass overwrites it
            });
        } else {
            // long horizons: consecutive passes with the same slot count share one rolled loop whose body is straight-line
            // (see PassGroups): the code stays small (one set of chains per DISTINCT slot count, not per pass)
            static_for<0, PassGroups<D>::count()>([&](auto gc) __attribute__((always_inline)) {
                constexpr PassGroups<D> pg{};
                constexpr int g = decltype(gc)::value;
#pragma unroll 1
                for (int m = pg.start[g]; m < pg.end[g]; ++m) {
                    VS_TIC();
                    rec_pass(m, pg.nks[g] == 9 ? 2 : 1, sY);
                    VS_TOC(0);
                    __syncthreads();
                    VS_TOC(1);
                    syrk_fixed(std::integral_constant<int, pg.nks[g]>{}, std::integral_constant<int, pg.nact[g]>{});
#ifdef VS_DIAG_PASS
                    if (m != VS_DIAG_PASS) { VS_TIC(); } else
#endif
                    VS_TOC(2);
                    __syncthreads();  // single Y buffer: the next pass overwrites it
                }
            });
        }
        VS_TIC();
    }
    if constexpr (!FUSED_DISPATCH) VS_STAMP(2);   // (the fused branch stamps its own P1 / P2 / P3 boundaries)
    VS_REFRESH_IDS();
    if constexpr (D::WG_PER_CU == 1) pin_tiles_agpr<TPW>(acc);   // long horizons: see pin_tiles_agpr

    // ---------------------------------------------------------------- P2 + P3 (wave-specialised, see cholesky_wave)
    constexpr int PVT = D::PVT;  // first tile row that contains a throttle row
    const int crow = (lane >> 4) * 17 + (lane & 15);  // C/D fragment: row (lane>>4)+4r, column lane&15
    const int lrow = (lane & 15) * 17 + (lane >> 4);  // A/B fragment: row lane&15, k = lane>>4
    // The debug dumps (condensed Hessian, factor) exist in the diagnostic instantiation only (STAMPS; the launcher picks it
    // when a dump or the stamps are asked for): in the shipped kernel their code -- input_cost_term per element, a branch
    // around every store -- cost registers at the joins for nothing.
    double* dbgM = STAMPS ? late_args()->dbgM : nullptr;
    double* dbgL = STAMPS ? late_args()->dbgL : nullptr;
    double* dbgLi = dbgL != nullptr ? dbgL + size_t(inst) * D::NP * D::NP : nullptr;
    if (STAMPS && !FUSED_DISPATCH && dbgM != nullptr) {  // debug/parity only: the augmented condensed Hessian before factorisation, from registers
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            if (kTileTab<D, PIPE>.forms(q * D::NWAVES + wave, wave)) {
                const int ti_q = kTileTab<D, PIPE>.ti[q * D::NWAVES + wave], tj_q = kTileTab<D, PIPE>.tj[q * D::NWAVES + wave];
                d4 tmp = acc[q];
                if (ti_q == tj_q || ti_q >= PVT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        tmp[r] += input_cost_term<D>(sCfg, smem + S::oQR + S::QR_GY, sVprev, 16 * ti_q + (lane >> 4) + 4 * r,
                                                     16 * tj_q + (lane & 15));
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gr = 16 * ti_q + (lane >> 4) + 4 * r, gc = 16 * tj_q + (lane & 15);
                    if (gc <= gr) dbgM[size_t(inst) * D::NP * D::NP + size_t(gr) * D::NP + gc] = tmp[r];
                }
            }
        }
    }
    if constexpr (!FUSED_DISPATCH) VS_STAMP(3);
    VS_REFRESH_IDS();
    if constexpr (!FUSED_DISPATCH)
    switch (wave) {  // scalar dispatch: every wavefront runs its own straight-line copy, same barrier count
        case 0: cholesky_wave<D, TPW, 0, STAMPS, PLDS, PIPE>(sCfg, acc, Lb, sInvD, smem + S::oQR + S::QR_GY, sVprev, sFlags, sXinv, sW, dbgLi, lane, crow, lrow, sZ); break;
        case 1: cholesky_wave<D, TPW, 1, STAMPS, PLDS, PIPE>(sCfg, acc, Lb, sInvD, smem + S::oQR + S::QR_GY, sVprev, sFlags, sXinv, sW, dbgLi, lane, crow, lrow, sZ); break;
        case 2: cholesky_wave<D, TPW, 2, STAMPS, PLDS, PIPE>(sCfg, acc, Lb, sInvD, smem + S::oQR + S::QR_GY, sVprev, sFlags, sXinv, sW, dbgLi, lane, crow, lrow, sZ); break;
        default: cholesky_wave<D, TPW, 3, STAMPS, PLDS, PIPE>(sCfg, acc, Lb, sInvD, smem + S::oQR + S::QR_GY, sVprev, sFlags, sXinv, sW, dbgLi, lane, crow, lrow, sZ); break;
    }
    static_assert(D::NWAVES == 4, "wave-specialised phases are instantiated for four wavefronts");
    if (STAMPS && dbgLi != nullptr) {  // debug/parity only: the factor; diagonal tiles were written while they were panels
        if constexpr (!FUSED_DISPATCH) {
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                const int ti_q = kTileTab<D, PIPE>.ti[q * D::NWAVES + wave], tj_q = kTileTab<D, PIPE>.tj[q * D::NWAVES + wave];
                if (kTileTab<D, PIPE>.holds(q * D::NWAVES + wave, wave) && tj_q < PVT && ti_q > tj_q) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dbgLi[size_t(16 * ti_q + (lane >> 4) + 4 * r) * D::NP + 16 * tj_q + (lane & 15)] = acc[q][r];
                }
            }
        }
        for (int e = tid; e < (D::NP - 16 * PVT) * (D::NP - 16 * PVT); e += D::BLOCK) {  // throttle corner, from LDS
            const int gr = 16 * PVT + e / (D::NP - 16 * PVT), gc = 16 * PVT + e % (D::NP - 16 * PVT);
            if (gc <= gr) dbgLi[size_t(gr) * D::NP + gc] = Lb[lower_at<D>(gr, gc)];
        }
    }

    VS_STAMP(4);
    VS_REFRESH_IDS();
    if constexpr (D::WG_PER_CU == 1) pin_tiles_agpr<TPW>(acc);
    // ---------------------------------------------------------------- P4/P5 back-substitution L^T z = y
    // Row NZ of the factor holds L^-1 g, so y = -row.  The throttles sit at the end of the order, hence the
    // first tiles of the backward sweep yield the throttles of the QP with only the hold pin enforced.  If
    // they respect their box the sweep simply continues into the joints (active-set iteration 1 of the
    // oracle's rule); otherwise the box QP on the Schur complement runs and the sweep restarts with the
    // throttles prescribed.
    // The joint columns of the factor are in registers by now (their right-hand side entries were put into sW when
    // the tiles came back from the panel); the throttle corner is in LDS, and so is everything the box QP touches.
    const bool hold = sIn[VSMPC_IN_HOLD] != 0.0;
    constexpr int PV = D::PVT;  // first tile that contains a throttle row
    auto init_corner_rhs = [&]() {
        if (tid >= 16 * PV && tid < D::NP) sW[tid] = tid < D::NZ ? -Lb[lower_at<D>(D::NZ, tid)] : 0.0;
    };
    init_corner_rhs();
    if (tid < D::NP) sZ[tid] = (hold && tid >= D::NZ - 4 && tid < D::NZ) ? sVprev[tid - (D::NZ - 4)] : 0.0;
    __syncthreads();

    // one tile step of the sweep over the corner; `prescribed` = throttles already fixed in sZ
    auto sweep_tile = [&](int p, bool prescribed) {
        if (wave == 0) {
            const int j = lane & 15;  // lane j owns column j of L_pp
            const int gj = 16 * p + j;
            const double* Tpp = Lb + tile_off<D>(p, p) + j;
            double colv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {  // column j of L_pp; above the diagonal the tile holds leftovers
                const double t = Tpp[k * 17];
                colv[k] = k >= j ? t : 0.0;
            }
            double w = sW[gj];
            // z_j = w_j * inv_eff + zadd: solved rows use 1/L_jj, prescribed rows (pinned or already fixed
            // throttles, gradient row, padding) use inv_eff = 0 and their value; branch-free in the chain
            const bool fix = (gj >= D::NZ) || (gj >= D::NU && (prescribed || (hold && gj >= D::NZ - 4)));
            const double inv_eff = fix ? 0.0 : sInvD[gj];
            const double zadd = (fix && gj < D::NZ) ? sZ[gj] : 0.0;
            double z = 0.0;
#pragma unroll
            for (int k = 15; k >= 0; --k) {
                const double zk = readlane_f64(fma(w, inv_eff, zadd), k);
                z = (j == k) ? zk : z;
                w = fma(-colv[k], zk, w);  // lanes j >= k: colv is zero or w is no longer used
            }
            if (lane < 16) sZ[gj] = z;
        }
        __syncthreads();
        if (p > PV) {  // corner columns only: the joint columns are updated from registers in P5
            if (tid >= 16 * PV && tid < 16 * p) {
                const double* T = Lb + tile_off<D>(p, tid >> 4) + (tid & 15);
                const double* zp = sZ + 16 * p;
                double a2 = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) a2 += T[k * 17] * zp[k];
                sW[tid] -= a2;
            }
            __syncthreads();
        }
    };

    constexpr bool DUALQP = S::DUALQP;
    if constexpr (DUALQP) {
        // The throttle block spans two tile rows.  Last tile row: only its KL = NZ - 16 (NT-1) throttle rows take part
        // (gradient row and padding have z = 0), the hold pins sit here.  First throttle tile row: no pins, and the
        // inverse of its diagonal tile is at hand (X66 from P3), so z = X66^T w needs no chain.
        constexpr int PL = D::NT - 1, KL = D::NZ - 16 * PL;
        static_assert(KL == D::NV - 16 && KL >= 4, "pins live in the last tile row");
        const double* X6 = sXinv + PV * D::TS;
        const double* L76 = Lb + tile_off<D>(PL, PV);
        const double* L77 = Lb + tile_off<D>(PL, PL);
        if (wave == 0) {
            // wavefront 0 runs both throttle tile rows back to back in registers (cross-lane traffic through
            // v_readlane only)
            const int j = lane & 15;
            const int gj = 16 * PL + j;
            const double* Tpp = L77 + j;
            double colv[KL], l76[KL], x6[16];
#pragma unroll
            for (int k = 0; k < KL; ++k) {  // column j of L_pp; above the diagonal the tile holds leftovers
                const double t = Tpp[k * 17];
                colv[k] = k >= j ? t : 0.0;
                l76[k] = L76[k * 17 + j];   // L[16 PL + k][16 PV + j]
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) x6[i] = X6[i * 17 + j];
            double w = sW[gj];
            double w6 = sW[16 * PV + j];
            const bool fix = (j >= KL) || (hold && j >= KL - 4);
            const double inv_eff = fix ? 0.0 : sInvD[gj];
            const double zadd = (fix && j < KL) ? sZ[gj] : 0.0;
            double z = 0.0;
#pragma unroll
            for (int k = KL - 1; k >= 0; --k) {
                const double zk = readlane_f64(fma(w, inv_eff, zadd), k);
                z = (j == k) ? zk : z;
                w = fma(-colv[k], zk, w);
                w6 = fma(-l76[k], zk, w6);  // right-hand side of the first throttle tile row, lane = row
            }
            double z6 = 0.0;                // z = X66^T w6: lane j sums X66[i][j] w6[i]
#pragma unroll
            for (int i = 0; i < 16; ++i) z6 = fma(x6[i], readlane_f64(w6, i), z6);
            if (lane < 16) { sZ[gj] = z; sZ[16 * PV + j] = z6; }
        }
        // (no barrier here: the only reader of these throttles before the next barrier is the violation check below, in this
        // same wavefront -- LDS operations of one wavefront execute in order)
    } else if constexpr (S::DUAL3) {
        // three throttle tile rows.  Wavefront 1 forms the inverse of the last corner diagonal tile for the box QP beside
        // wavefront 0's sweep (see cholesky_wave for the second one), a few rows in front of every step so that no barrier
        // of the sweep waits for it
        static_assert(D::NT - 1 == PV + 2, "three throttle tile rows");
        double x2[16];
        const double* L22d = Lb + tile_off_c<D>(PV + 2, PV + 2);
        const double* inv2 = sInvD + D::NU + 32;
        if (wave == 1) tile_inverse_rows<0, 8>(L22d, inv2, x2, lane);
        sweep_tile(PV + 2, false);
        // the other two tile rows have no pinned or prescribed entry in this pass and their inverses are at hand (X_PVT from
        // P3, the second one from P3's last panel): z = X^T w, sixteen multiply-adds per lane instead of a 16-step
        // broadcast chain
        auto sweep_tile_x = [&](int p, const double* Xp) {
            if (wave == 0) {
                const int j = lane & 15;
                double z0 = 0.0, z1 = 0.0;
#pragma unroll
                for (int k = 0; k < 16; k += 2) {   // X[k][j] = 0 for k < j (stored zeros); w: uniform addresses
                    z0 = fma(Xp[k * 17 + j], sW[16 * p + k], z0);
                    z1 = fma(Xp[(k + 1) * 17 + j], sW[16 * p + k + 1], z1);
                }
                if (lane < 16) sZ[16 * p + j] = z0 + z1;
            }
            __syncthreads();
            if (p > PV) {
                if (tid >= 16 * PV && tid < 16 * p) {
                    const double* T = Lb + tile_off<D>(p, tid >> 4) + (tid & 15);
                    const double* zp = sZ + 16 * p;
                    double a2 = 0.0;
#pragma unroll
                    for (int k = 0; k < 16; ++k) a2 += T[k * 17] * zp[k];
                    sW[tid] -= a2;
                }
                __syncthreads();
            }
        };
        if (wave == 1) tile_inverse_rows<8, 13>(L22d, inv2, x2, lane);
        sweep_tile_x(PV + 1, smem + S::oDual3T0);
        if (wave == 1) {
            tile_inverse_rows<13, 16>(L22d, inv2, x2, lane);
            if (lane < 16) {
#pragma unroll
                for (int i = 0; i < 16; ++i) smem[S::oDual3T1 + i * 17 + lane] = x2[i];
            }
        }
        sweep_tile_x(PV, sXinv + PV * D::TS);
    } else {
#pragma unroll 1
        for (int p = D::NT - 1; p >= PV; --p) sweep_tile(p, false);
    }
    if (wave == 0) {
        const bool valid = lane < D::NV;
        const double v = sZ[D::NU + (valid ? lane : 0)];
        const bool fixed = hold && lane >= D::NV - 4;
        const double tolv = 1e-12 * (1.0 + fabs(v));
        const bool viol = valid && !fixed && (v < sCfg[CFG_VMIN] - tolv || v > sCfg[CFG_VMAX] + tolv);
        const unsigned long long vm = __ballot(viol);
        if (lane == 0) { sFlags[3] = __popcll(vm); sFlags[1] = VSMPC_STATUS_SOLVED; sFlags[2] = 1; }
    }
    __syncthreads();
    const bool need_qp = sFlags[3] != 0;
    VS_STAMP(5);
    VS_REFRESH_IDS();

    if (need_qp) {
        box_qp<D>(sFlags[3], hold, wave);
        __syncthreads();
        VS_STAMP(6);
        VS_REFRESH_IDS();
        if constexpr (D::NU % 16 != 0) {
            // joint rows share the first corner tile row with throttle rows: redo the corner sweep with the throttles
            // prescribed (its right-hand side starts over from y)
            init_corner_rhs();
            __syncthreads();
#pragma unroll 1
            for (int p = D::NT - 1; p >= PV; --p) sweep_tile(p, true);
        }
    } else {
        VS_STAMP(6);
    }
    // ---------------------------------------------------------------- P5 joints from the register-resident factor
    switch (wave) {
        case 0:
            if constexpr (PIPE) p5_wave0_jets<D>(smem, lane);
            else backsub_wave<D, TPW, 0, PIPE>(acc, sW, sZ, sXinv, sU, lane);
            break;
        case 1: backsub_wave<D, TPW, 1, PIPE>(acc, sW, sZ, sXinv, sU, lane); break;
        case 2: backsub_wave<D, TPW, 2, PIPE>(acc, sW, sZ, sXinv, sU, lane); break;
        default: backsub_wave<D, TPW, 3, PIPE>(acc, sW, sZ, sXinv, sU, lane); break;
    }
    // (the throttles were final before P5: their copy needs no barrier in front of it; the one behind it also covers the joints)
    if (tid < D::NV) sV[tid] = sZ[D::NU + tid];
    __syncthreads();

    VS_STAMP(7);
    VS_REFRESH_IDS();
    // ---------------------------------------------------------------- P6 forward simulation + outputs
    // the output pointers: requested from the kernarg segment here, a phase ahead of their use (a scalar load is a
    // ~0.5 us round trip when it misses, and nothing else in P6 wants the scalar registers)
#ifdef VS_DIAG_P6   // measurement build: P6 in four pieces (input terms / set-up + first step / other steps / outputs) -> t_acc[0..3]
    for (int i = 0; i < 4; ++i) t_acc[i] = 0;
    VS_TIC();
#define VS_P6_TOC(i) VS_TOC(i)
#else
#define VS_P6_TOC(i) do { } while (0)
#endif
    const SolveArgs* ka = late_args();
    double* xout = ka->xout;
    double* fmout = ka->fmout;
    int* status_out = ka->status_out;
    int* iters_out = ka->iters_out;
    // wavefront 3 has no element of the input-term pass below (pipelined schedule: 6 N of them): it starts taking the reduced joint
    // unknowns back to joint increments here -- the first three reflectors; the rest beside the first pipeline step
    double u8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    constexpr int JX_SPLIT = PIPE ? 3 : NJC;
    if (PIPE && wave == 3 && lane < D::HC) joint_expand<D, NJC, JX_SPLIT>(smem + S::oQR, sZ + NJC * lane, u8);
    // input terms of every stage in parallel: f_k = Bj U_{jb(k)} + Bt v_{tb(k)} + c  (Bj U = R^T y in the reduced unknowns)
    // (straight-line rounds with clamped indices: the loads of all rounds are in flight together; as a loop with a per-thread
    // trip count the rounds ran one LDS round trip after the other)
    // Pipelined schedule: only the six momentum rows are read from here (the jets formed theirs beside P5, the CoM / RPY link has
    // constant input terms and reads c itself): 6 N elements, one round.
    {
        constexpr int ROWS = PIPE ? 6 : NX;
        constexpr int NE = ROWS * D::N, RND = (NE + D::BLOCK - 1) / D::BLOCK;
#pragma unroll
        for (int rd = 0; rd < RND; ++rd) {
            const int e0 = tid + rd * D::BLOCK, ec = e0 < NE ? e0 : NE - 1;
            const int k = ec / ROWS, rr = ec - k * ROWS;
            const int r = PIPE ? (rr < 3 ? 3 + rr : 6 + rr) : rr;   // rows 3..5, 9..11
            const int e = NX * k + r;
            const int jb = joint_block_of_stage<D>(k);
            const int tb = throttle_block_of_stage<D>(k);
            const int vq = tb == 0 ? D::NV - 4 : 4 * (tb - 1);  // internal offset of reference block tb
            double f = sC[r];
#pragma unroll
            for (int c = 0; c < NJC; ++c) f += sBj[r * NJ + c] * sZ[NJC * jb + c];
#pragma unroll
            for (int c = 0; c < NTH; ++c) f += sBt[r * NTH + c] * sV[vq + c];
            if (e0 < NE) sF[e] = f;
        }
    }
    __syncthreads();
    VS_P6_TOC(0);
    {
        // The three links of the cascade (jets -> momenta -> CoM / RPY + error integrators; systemDynamicsVSMPC.cpp:
        // 79-103,288-319,384-429) run in THREE wavefronts, one chunk of CHK stages apart: step s = jets of chunk s (wavefront
        // 0), momenta of chunk s - 1 (wavefront 1, after adding A_mom T_k to its forcing), CoM / RPY of chunk s - 2
        // (wavefront 2); a workgroup barrier per step hands the trajectories over through sX.  NCH + 2 steps instead of the
        // 3 NCH chunk-lengths one wavefront needs for the links in series (through v19: 8.1 k cycles at the paper horizon,
        // 17.0 k at the 2x horizon; now 7.1 k / 11.6 k).
#ifndef VS_P6_CHK
#define VS_P6_CHK (D::N > 20 ? 9 : 6)
#endif
        constexpr int CHK = VS_P6_CHK, NCH = (D::N + CHK - 1) / CHK;
        // chain states (registers of the owning lanes, alive across the steps)
        double jT = 0.0, jTd = 0.0, jon = 0.0, ja = 0.0, jb = 0.0;
        double Sk[9], hh[3] = {0.0, 0.0, 0.0};
        double cm0 = 0.0, cm1 = 0.0, cm2 = 0.0, cce = 0.0, cx = 0.0, cee = 0.0;
        const int hr0m = (lane & 1) ? 9 : 3;                       // wavefront 1, lane < 2: h_lin / h_ang
        const int cg = lane / 3, cr = lane - 3 * cg;               // wavefront 2, lane < 6: (half, row)
        const int cxr = (cg ? 6 : 0) + cr, chr0 = cg ? 9 : 3, cer = (cg ? 23 : 20) + cr;
        // (pipelined schedule: the jets ran beside P5, p5_wave0_jets; the cascade starts at the momenta)
        constexpr bool JETS_EARLY = PIPE;
        constexpr int JOFF = JETS_EARLY ? 0 : 1;
        if (!JETS_EARLY && wave == 0 && lane < NTH) {
            jon = sA[(12 + lane) * NX + 16 + lane]; ja = sA[(16 + lane) * NX + 12 + lane]; jb = sA[(16 + lane) * NX + 16 + lane];
            jT = sIn[VSMPC_IN_X0 + 12 + lane]; jTd = sIn[VSMPC_IN_X0 + 16 + lane];
            sX[12 + lane] = jT;
            sX[16 + lane] = jTd;
        }
        if (wave == 1 && lane < 2) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                hh[r] = sIn[VSMPC_IN_X0 + hr0m + r];
                sX[hr0m + r] = hh[r];
#pragma unroll
                for (int c = 0; c < 3; ++c) Sk[3 * r + c] = sA[(hr0m + r) * NX + hr0m + c];
            }
        }
        if (wave == 2 && lane < 6) {
            cm0 = sA[cxr * NX + chr0]; cm1 = sA[cxr * NX + chr0 + 1]; cm2 = sA[cxr * NX + chr0 + 2];
            cce = sC[cer];
            cx = sIn[VSMPC_IN_X0 + cxr]; cee = sIn[VSMPC_IN_X0 + cer];
            sX[cxr] = cx;
            sX[cer] = cee;
        }
        // wavefront 3 (no link of the cascade) takes the reduced joint unknowns back to joint increments meanwhile:
        // U_i = W^(-1/2) (Q y_i + N n), one block per lane, into the (dead) partial-sum array of P5
        static_assert(D::NUO <= D::NWAVES * D::NP, "joint increments fit the partial-sum array");
        if (wave == 3 && lane < D::HC) {
            if constexpr (PIPE) joint_expand<D, JX_SPLIT, 0>(smem + S::oQR, sZ + NJC * lane, u8);
            else joint_expand<D>(smem + S::oQR, sZ + NJC * lane, u8);
#pragma unroll
            for (int i = 0; i < 8; ++i) sU[NJ * lane + i] = u8[i];
        }
        // ... and sends the input part of the primal on its way to HBM while the cascade runs (the state part follows at the end)
        if (wave == 3 && xout != nullptr) {
            double2* xo = reinterpret_cast<double2*>(xout + size_t(inst) * D::NVAR);  // 16 B per lane stores
            for (int i = lane; i < D::NUO / 2; i += 64) xo[D::NXS / 2 + i] = make_double2(sU[2 * i], sU[2 * i + 1]);
            if (lane < D::NV / 2) {  // reference order v_0..v_{NVB-1}
                const int e = 2 * lane, b = e >> 2, c = e & 3;
                const int q = b == 0 ? D::NV - 4 + c : 4 * (b - 1) + c;
                xo[(D::NXS + D::NUO) / 2 + lane] = make_double2(sV[q], sV[q + 1]);
            }
        }

        static_for<0, NCH + 1 + JOFF>([&](auto scst) __attribute__((always_inline)) {
            constexpr int st = decltype(scst)::value;
            if constexpr (!JETS_EARLY && st < NCH) {   // jets, chunk st
                if (wave == 0 && lane < NTH) {
                    constexpr int k0 = st * CHK;
                    double fa[CHK], fb[CHK], dtk[CHK];
#pragma unroll
                    for (int u = 0; u < CHK; ++u) {
                        const int k = (k0 + u < D::N) ? k0 + u : D::N - 1;
                        fa[u] = sF[NX * k + 12 + lane];
                        fb[u] = sF[NX * k + 16 + lane];
                        dtk[u] = sDt[k];
                    }
#pragma unroll
                    for (int u = 0; u < CHK; ++u) {
                        const double dT = fma(jon, jTd, fa[u]);
                        const double dTd = fma(ja, jT, fma(jb, jTd, fb[u]));
                        jT = fma(dtk[u], dT, jT);
                        jTd = fma(dtk[u], dTd, jTd);
                        if (k0 + u < D::N) {
                            sX[NX * (k0 + u + 1) + 12 + lane] = jT;
                            sX[NX * (k0 + u + 1) + 16 + lane] = jTd;
                        }
                    }
                }
            }
            // pipelined schedule: wavefront 3 sends the first-move block beside the second step -- everything in it is known (the
            // jets' node 1 since P5), and its throttle percentages cost a square root that has no business on the kernel's last stretch
            if constexpr (PIPE && st == 1) {
                if (wave == 3 && fmout != nullptr && lane < VSMPC_FM_SIZE) {
                    double v;
                    if (lane < 8) v = sU[lane];                                        // delta q           (variableSamplingMPC.cpp:99)
                    else if (lane < 12) v = sV[D::NV - 4 + (lane - 8)];                // v0                (:100)
                    else if (lane < 16) v = Jet::throttle_of_v(sV[D::NV - 4 + (lane - 12)]);  // throttle % (:146-149)
                    else if (lane < 20) v = sX[NX + 12 + (lane - 16)];                 // thrust, node 1    (:101)
                    else v = sX[NX + 16 + (lane - 20)];                                // thrust rate, node 1 (:102)
                    fmout[size_t(inst) * VSMPC_FM_SIZE + lane] = v;
                }
            }
            if constexpr (st >= JOFF && st - JOFF < NCH) {   // momenta, chunk st - JOFF
                if (wave == 1) {
                    constexpr int k0 = (st - JOFF) * CHK;
                    constexpr int kn = k0 + CHK < D::N ? CHK : D::N - k0;   // stages of this chunk
                    // forcing g_k = A_mom T_k + f_k on the six momentum rows of the chunk's stages (T_k: the previous step's jets)
                    if (lane < 6 * kn) {
                        const int k = k0 + lane / 6, rr = lane % 6, row = rr < 3 ? 3 + rr : 6 + rr;   // rows 3..5, 9..11
                        double gk = sF[NX * k + row];
#pragma unroll
                        for (int c = 0; c < NTH; ++c) gk = fma(sA[row * NX + 12 + c], sX[NX * k + 12 + c], gk);
                        sF[NX * k + row] = gk;
                    }
                    if (lane < 2) {   // (LDS operations of one wavefront execute in order: the forcing above is visible)
                        double gk[CHK][3], dtk[CHK];
#pragma unroll
                        for (int u = 0; u < CHK; ++u) {
                            const int k = (k0 + u < D::N) ? k0 + u : D::N - 1;
#pragma unroll
                            for (int r = 0; r < 3; ++r) gk[u][r] = sF[NX * k + hr0m + r];
                            dtk[u] = sDt[k];
                        }
#pragma unroll
                        for (int u = 0; u < CHK; ++u) {
                            double dh[3];
#pragma unroll
                            for (int r = 0; r < 3; ++r)
                                dh[r] = fma(Sk[3 * r], hh[0], fma(Sk[3 * r + 1], hh[1], fma(Sk[3 * r + 2], hh[2], gk[u][r])));
#pragma unroll
                            for (int r = 0; r < 3; ++r) hh[r] = fma(dtk[u], dh[r], hh[r]);
                            if (k0 + u < D::N) {
#pragma unroll
                                for (int r = 0; r < 3; ++r) sX[NX * (k0 + u + 1) + hr0m + r] = hh[r];
                            }
                        }
                    }
                }
            }
            if constexpr (st >= JOFF + 1 && st - JOFF - 1 < NCH) {   // CoM / RPY and their error integrators, chunk st - JOFF - 1
                if (wave == 2 && lane < 6) {
                    constexpr int k0 = (st - JOFF - 1) * CHK;
                    double hk[CHK][3], dtk[CHK];
#pragma unroll
                    for (int u = 0; u < CHK; ++u) {
                        const int k = (k0 + u < D::N) ? k0 + u : D::N - 1;
#pragma unroll
                        for (int c = 0; c < 3; ++c) hk[u][c] = sX[NX * k + chr0 + c];
                        dtk[u] = sDt[k];
                    }
#pragma unroll
                    for (int u = 0; u < CHK; ++u) {
                        const double dx = fma(cm0, hk[u][0], fma(cm1, hk[u][1], cm2 * hk[u][2]));
                        cee = fma(dtk[u], cx + cce, cee);
                        cx = fma(dtk[u], dx, cx);
                        if (k0 + u < D::N) {
                            sX[NX * (k0 + u + 1) + cxr] = cx;
                            sX[NX * (k0 + u + 1) + cer] = cee;
                        }
                    }
                }
            }
            if constexpr (st + 1 < NCH + 1 + JOFF) __syncthreads();
            if constexpr (st == 0) VS_P6_TOC(1);
        });
    }
    __syncthreads();
    VS_P6_TOC(2);
    VS_STAMP(8);
    VS_REFRESH_IDS();

    if (xout != nullptr) {
        double2* xo = reinterpret_cast<double2*>(xout + size_t(inst) * D::NVAR);  // 16 B per lane stores
        for (int i = tid; i < D::NXS / 2; i += D::BLOCK) xo[i] = make_double2(sX[2 * i], sX[2 * i + 1]);
        // (the joint increments and the throttles left from wavefront 3 during the cascade)
    }
    if (!PIPE && fmout != nullptr && tid < VSMPC_FM_SIZE) {   // (pipelined schedule: sent by wavefront 3 during the cascade)
        double v;
        if (tid < 8) v = sU[tid];                                        // delta q           (variableSamplingMPC.cpp:99)
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
    VS_P6_TOC(3);
    VS_STAMP(9);
    if constexpr (STAMPS) {
        unsigned long long* st_ = late_args()->stamps;
        if (tid == 0 && st_ != nullptr) {
#if defined(VS_DIAG_P3) && !defined(VS_DIAG_P6)
            for (int i = 0; i < 4; ++i) t_acc[i] = vs_diag_p3[i];
#endif
            t_acc[5] = __builtin_amdgcn_s_memrealtime() - rt0;
            t_acc[4] = rt0;  // absolute start (global 100 MHz counter): start skew across the workgroups of a launch
            for (int i = 0; i < 6; ++i) st_[size_t(blockIdx.x) * 16 + 10 + i] = t_acc[i];
        }
    }
#undef VS_STAMP
#undef VS_REFRESH_IDS
#undef VS_TIC
#undef VS_TOC
}

#if !defined(VS_TU_HORIZON)
// ------------------------------------------------------------------------------------------------
// kinematics-derived inputs (vsmpc_kinematics_batch): Lambda_lin,B, Lambda_ang,B, I_G per instance.
// HBM-bound (5.6 KB in, 0.46 KB out per instance): one wavefront per instance stages the record in LDS with
// 16 B/lane loads, 57 lanes compute one output element each.
//   computeLambdaLin          systemDynamicsVSMPC.cpp:321-350
//   computeLambdaAng          systemDynamicsVSMPC.cpp:159-206 ("unfiltered"), getRelativeJacobianCoM :208-226
//   locked inertia I_G        systemDynamicsVSMPC.cpp:128-130 (iDynTree adjoint X = [R, S(r)R; 0, R])
// ------------------------------------------------------------------------------------------------
// Options (vsmpc_set_kinematics_options): `sel` = robot joint index of each controlled joint for Lambda_ang, which the
// reference selects by NAME (systemDynamicsVSMPC.cpp:57-66,202-205; Lambda_lin keeps the reference's hard-coded column
// offset 3, :348); `constant_lambda` = jointsLambdaOption "constant" (:186-200,329-337): axes, arms and relative Jacobians
// are the configure-time ones, the angular term uses the relative Jacobian's own top rows (delivered in the JFRAME slot)
// instead of R^T (J_frame - J_CoM) and the thrusts of getRobot() (delivered in JCOM[0..3]).
__global__ __launch_bounds__(64) void kinematics_kernel(const double* __restrict__ kin, int batch,
                                                        double* __restrict__ out, KinOpts opts) {
    __shared__ __attribute__((aligned(16))) double s[VSMPC_KIN_SIZE + 1];
    __shared__ double sRa[12], sRr[12];  // R^T a_i, R^T r_i
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= batch) return;
    const double* rec = kin + size_t(b) * VSMPC_KIN_SIZE;
    // the record stride (697 doubles) is odd, so 16 B alignment alternates: peel one element where needed
    const int head = (reinterpret_cast<size_t>(rec) & 15) ? 1 : 0;
    if (lane == 0 && head) s[0] = rec[0];
    const double2* r2 = reinterpret_cast<const double2*>(rec + head);
    const int n2 = (VSMPC_KIN_SIZE - head) / 2;
    for (int i = lane; i < n2; i += 64) {
        const double2 v = r2[i];
        s[head + 2 * i] = v.x;
        s[head + 2 * i + 1] = v.y;
    }
    if (lane == 0 && ((VSMPC_KIN_SIZE - head) & 1)) s[VSMPC_KIN_SIZE - 1] = rec[VSMPC_KIN_SIZE - 1];
    __syncthreads();
    const double* R = s + VSMPC_KIN_WRB;
    if (lane < 24) {  // R^T a_i and R^T r_i
        const int which = lane / 12, e = lane % 12, i = e / 3, c = e % 3;
        const double* v = s + (which ? VSMPC_KIN_ARMS : VSMPC_KIN_AXES) + 3 * i;
        const double val = R[c] * v[0] + R[3 + c] * v[1] + R[6 + c] * v[2];
        (which ? sRr : sRa)[e] = val;
    }
    __syncthreads();
    constexpr int NJq = VSMPC_KIN_NJ, OFF = 3;  // Lambda_lin: robot joints 3..10 (systemDynamicsVSMPC.cpp:348)
    double res = 0.0;
    if (lane < 48) {
        const bool ang = lane >= 24;
        const int e = lane % 24, r = e >> 3, col = ang ? opts.sel[e & 7] : OFF + (e & 7);
        for (int i = 0; i < 4; ++i) {
            const double T = (ang && opts.constant_lambda) ? s[VSMPC_KIN_JCOM + i] : s[VSMPC_KIN_THRUST + i];
            const double* a = sRa + 3 * i;
            const double* Jrel = s + VSMPC_KIN_JREL + i * 3 * NJq;
            // w = S(a) * Jrel[:, col]  (skew: FlightControlUtils.cpp:77-85)
            const double j0 = Jrel[col], j1 = Jrel[NJq + col], j2 = Jrel[2 * NJq + col];
            const double w0 = -a[2] * j1 + a[1] * j2, w1 = a[2] * j0 - a[0] * j2, w2 = -a[1] * j0 + a[0] * j1;
            if (!ang) {
                res -= T * (r == 0 ? w0 : (r == 1 ? w1 : w2));
            } else {
                const double* Jf = s + VSMPC_KIN_JFRAME + i * 3 * NJq;
                const double* Jc = s + VSMPC_KIN_JCOM;
                const double d0 = Jf[col] - Jc[col], d1 = Jf[NJq + col] - Jc[NJq + col], d2 = Jf[2 * NJq + col] - Jc[2 * NJq + col];
                double g0 = R[0] * d0 + R[3] * d1 + R[6] * d2;  // R^T (J_frame - J_CoM)
                double g1 = R[1] * d0 + R[4] * d1 + R[7] * d2;
                double g2 = R[2] * d0 + R[5] * d1 + R[8] * d2;
                if (opts.constant_lambda) { g0 = Jf[col]; g1 = Jf[NJq + col]; g2 = Jf[2 * NJq + col]; }
                const double u0 = -a[2] * g1 + a[1] * g2, u1 = a[2] * g0 - a[0] * g2, u2 = -a[1] * g0 + a[0] * g1;
                const double* q = sRr + 3 * i;  // S(R^T r_i) * w
                const double z0 = -q[2] * w1 + q[1] * w2, z1 = q[2] * w0 - q[0] * w2, z2 = -q[1] * w0 + q[0] * w1;
                res -= T * ((r == 0 ? u0 : (r == 1 ? u1 : u2)) + (r == 0 ? z0 : (r == 1 ? z1 : z2)));
            }
        }
    } else if (lane < 57) {
        // I_G = [S(r)R; R]^T M_b [S(r)R; R], element (i, j)
        const int e = lane - 48, i = e / 3, j = e % 3;
        const double* rr = s + VSMPC_KIN_R;
        const double* M = s + VSMPC_KIN_MB;
        double Xi[6], Xj[6];  // columns i and j of the 6x3 matrix [S(r)R; R]
        for (int k = 0; k < 3; ++k) { Xi[3 + k] = R[3 * k + i]; Xj[3 + k] = R[3 * k + j]; }
        Xi[0] = -rr[2] * Xi[4] + rr[1] * Xi[5]; Xi[1] = rr[2] * Xi[3] - rr[0] * Xi[5]; Xi[2] = -rr[1] * Xi[3] + rr[0] * Xi[4];
        Xj[0] = -rr[2] * Xj[4] + rr[1] * Xj[5]; Xj[1] = rr[2] * Xj[3] - rr[0] * Xj[5]; Xj[2] = -rr[1] * Xj[3] + rr[0] * Xj[4];
        for (int a = 0; a < 6; ++a) {
            double t = 0.0;
            for (int c = 0; c < 6; ++c) t += M[6 * a + c] * Xj[c];
            res += Xi[a] * t;
        }
    }
    if (out != nullptr && lane < VSMPC_KIN_OUT) out[size_t(b) * VSMPC_KIN_OUT + lane] = res;
    if (opts.records != nullptr && lane < (opts.skip_inertia ? 48 : VSMPC_KIN_OUT)) {   // device-resident input records: LLIN | LANG | INERTIA
        double* rec = opts.records + size_t(b) * opts.n_in;
        rec[(lane < 24 ? VSMPC_IN_LLIN + lane : (lane < 48 ? VSMPC_IN_LANG + lane - 24 : VSMPC_IN_INERTIA + lane - 48))] = res;
    }
}

hipError_t launch_kinematics_patch(const double* d_kin, int batch, double* d_records, int n_in, const KinOpts& opts,
                                   hipStream_t stream) {
    KinOpts o = opts;
    o.records = d_records;
    o.n_in = n_in;
    hipLaunchKernelGGL(kinematics_kernel, dim3(batch), dim3(64), 0, stream, d_kin, batch, static_cast<double*>(nullptr), o);
    return hipGetLastError();
}

hipError_t launch_kinematics(const double* d_kin, int batch, double* d_out, const KinOpts& opts, hipStream_t stream) {
    hipLaunchKernelGGL(kinematics_kernel, dim3(batch), dim3(64), 0, stream, d_kin, batch, d_out, opts);
    return hipGetLastError();
}

#endif  // !VS_TU_HORIZON (common part: the non-template kernels)

// ------------------------------------------------------------------------------------------------
// launchers.  The kernels are straight-line template instantiations over Dims<nIter, nIterSmall, controlHorizon>; the
// table of instantiated horizons is csrc/vsmpc_horizons.def (one X(...) line per horizon, generated by build.py from
// the list of horizons to support).  Variant ids are 1-based positions in that table.
//
// Translation units.  Every horizon is ~10 instantiations of a 30 k-instruction kernel, and a monolithic build compiled
// them one after the other (4.5 minutes).  build.py therefore compiles this file several times, in parallel:
//   -DVS_TU_COMMON                         the non-template kernels, the horizon table and the dispatchers
//   -DVS_TU_HORIZON=N,NS,HC -DVS_TU_STAMPS=0|1   the production (0) or diagnostic (1) solve kernels of ONE horizon
//                                          (+ its linearise kernel in the production unit), as explicit instantiations of
//                                          launch_solve_dims / launch_linearize_dims, which the dispatchers only declare
// Without either macro the file is one monolithic unit, as before.
// ------------------------------------------------------------------------------------------------
constexpr int MAX_DEVICES = 64;

template <int N, int NS, int HC, bool STAMPS>
hipError_t launch_solve_dims(int form, const DevCfg& cfg, const double* d_in, int batch, double* d_x, double* d_fm,
                             int* d_status, int* d_iters, double* dbgM, double* dbgL, unsigned long long* stamps,
                             hipStream_t stream);
template <int N, int NS, int HC>
hipError_t launch_linearize_dims(const DevCfg& cfg, const double* d_in, int batch, double* A, double* Bj, double* Bt,
                                 double* c, hipStream_t stream);

#if !defined(VS_TU_COMMON)
// ---- per-horizon part
template <class D, bool STAMPS, int FORM, bool PLDS = false>
static hipError_t launch_solve_f(int dev, const DevCfg& cfg, const double* d_in, int batch, double* d_x, double* d_fm,
                                 int* d_status, int* d_iters, double* dbgM, double* dbgL,
                                 unsigned long long* stamps, hipStream_t stream) {
    // the dynamic-LDS limit is a per-device function attribute: one process may drive several devices
    // (two host threads with their own handles may arrive here together: the flag is atomic, setting the attribute twice
    // is harmless, and it is published only after the call has succeeded)
    static std::atomic<bool> attr_set[MAX_DEVICES];
    constexpr size_t lds = FORM == 1 ? Smem<D>::bytes_struct : Smem<D>::bytes;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_kernel<D, STAMPS, FORM, PLDS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
        attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((solve_kernel<D, STAMPS, FORM, PLDS>), dim3(batch), dim3(D::BLOCK), lds, stream,
                       cfg, d_in, batch, d_x, d_fm, d_status, d_iters, dbgM, dbgL, stamps);
    return hipGetLastError();
}

template <int N, int NS, int HC, bool STAMPS>
hipError_t launch_solve_dims(int form, const DevCfg& cfg, const double* d_in, int batch, double* d_x, double* d_fm,
                             int* d_status, int* d_iters, double* dbgM, double* dbgL, unsigned long long* stamps,
                             hipStream_t stream) {
    using D = Dims<N, NS, HC>;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
    if constexpr (D::STRUCT_P1) {
        if (form != 2) {
            // two workgroups per CU once the batch exceeds the CUs: the panel streams with LDS broadcasts (see panel_factor)
            // (only the C++ streams have that form; the DPP streams broadcast inside the FMA)
            if constexpr (D::WG_PER_CU == 2 && !STAMPS && !VS_PANEL_DPP) {
                static std::atomic<int> cus[MAX_DEVICES];
                int ncu = cus[dev].load(std::memory_order_relaxed);
                if (ncu == 0) {
                    int n = 0;
                    e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
                    if (e != hipSuccess) return e;
                    ncu = n > 0 ? n : 1;
                    cus[dev].store(ncu, std::memory_order_relaxed);
                }
                const char* pv = getenv("VSMPC_PANEL");   // measurements: lds | readlane
                const bool lds = pv != nullptr ? pv[0] == 'l' : batch > ncu;
                if (lds)
                    return launch_solve_f<D, STAMPS, 1, true>(dev, cfg, d_in, batch, d_x, d_fm, d_status, d_iters, dbgM, dbgL,
                                                              stamps, stream);
            }
            return launch_solve_f<D, STAMPS, 1>(dev, cfg, d_in, batch, d_x, d_fm, d_status, d_iters, dbgM, dbgL, stamps,
                                                stream);
        }
    }
    return launch_solve_f<D, STAMPS, 0>(dev, cfg, d_in, batch, d_x, d_fm, d_status, d_iters, dbgM, dbgL, stamps,
                                            stream);
}

template <int N, int NS, int HC>
hipError_t launch_linearize_dims(const DevCfg& cfg, const double* d_in, int batch, double* A, double* Bj, double* Bt,
                                 double* c, hipStream_t stream) {
    hipLaunchKernelGGL((linearize_kernel<Dims<N, NS, HC>>), dim3(batch), dim3(256), 0, stream, cfg, d_in, A, Bj, Bt, c);
    return hipGetLastError();
}

#define VS_INSTANTIATE_SOLVE(N, NS, HC, ST)                                                                              \
    template hipError_t launch_solve_dims<N, NS, HC, ST>(int, const DevCfg&, const double*, int, double*, double*, int*, \
                                                         int*, double*, double*, unsigned long long*, hipStream_t);
#define VS_INSTANTIATE_LIN(N, NS, HC)                                                                                   \
    template hipError_t launch_linearize_dims<N, NS, HC>(const DevCfg&, const double*, int, double*, double*, double*, \
                                                         double*, hipStream_t);
#if defined(VS_TU_HORIZON)
#define VS_TU_APPLY(M, ...) M(__VA_ARGS__)
#if VS_TU_STAMPS
VS_TU_APPLY(VS_INSTANTIATE_SOLVE, VS_TU_HORIZON, true)
#else
VS_TU_APPLY(VS_INSTANTIATE_SOLVE, VS_TU_HORIZON, false)
VS_TU_APPLY(VS_INSTANTIATE_LIN, VS_TU_HORIZON)
#endif
#else   // monolithic unit: every horizon of the table
#define X(N, NS, HC) VS_INSTANTIATE_SOLVE(N, NS, HC, true) VS_INSTANTIATE_SOLVE(N, NS, HC, false) VS_INSTANTIATE_LIN(N, NS, HC)
#include "vsmpc_horizons.def"
#undef X
#endif
#endif  // per-horizon part

#if !defined(VS_TU_HORIZON)
// ---- common part
// condensing form of a handle: 0 = the default of the horizon (structured where Dims::STRUCT_P1), 1 = structured,
// 2 = SYRK (vsmpc_set_kernel_form).  VSMPC_FORM=structured|syrk is what a new handle starts with, for measurements of
// unmodified programs.
int initial_kernel_form() {
    const char* v = getenv("VSMPC_FORM");
    return (v != nullptr && v[0] == 's' && v[1] == 't') ? 1 : (v != nullptr && v[0] == 's' && v[1] == 'y') ? 2 : 0;
}

struct HorizonEntry {
    int n_iter, n_iter_small, control_horizon, n_p;
    size_t lds_bytes;
    const char* name;
    bool structured;
};
#define VSMPC_STR2(x) #x
#define VSMPC_STR(x) VSMPC_STR2(x)
static const HorizonEntry kHorizons[] = {
#define X(N, NS, HC) {N, NS, HC, Dims<N, NS, HC>::NP, Dims<N, NS, HC>::STRUCT_P1 ? Smem<Dims<N, NS, HC>>::bytes_struct : Smem<Dims<N, NS, HC>>::bytes, "solve_kernel<Dims<" VSMPC_STR(N) "," VSMPC_STR(NS) "," VSMPC_STR(HC) ">>", Dims<N, NS, HC>::STRUCT_P1},
#include "vsmpc_horizons.def"
#undef X
};
constexpr int kNumHorizons = int(sizeof(kHorizons) / sizeof(kHorizons[0]));

int select_variant(int n_iter, int n_iter_small, int control_horizon) {
    for (int i = 0; i < kNumHorizons; ++i)
        if (kHorizons[i].n_iter == n_iter && kHorizons[i].n_iter_small == n_iter_small &&
            kHorizons[i].control_horizon == control_horizon)
            return i + 1;
    return VARIANT_NONE;
}

int num_variants() { return kNumHorizons; }

void variant_horizon(int variant, int* n_iter, int* n_iter_small, int* control_horizon) {
    const HorizonEntry& h = kHorizons[variant - 1];
    *n_iter = h.n_iter;
    *n_iter_small = h.n_iter_small;
    *control_horizon = h.control_horizon;
}

const char* variant_kernel_name(int variant) {
    return (variant >= 1 && variant <= kNumHorizons) ? kHorizons[variant - 1].name : "none";
}

int variant_condensed_dim(int variant) {
    return (variant >= 1 && variant <= kNumHorizons) ? kHorizons[variant - 1].n_p : 0;
}

bool variant_has_structured(int variant) {
    return variant >= 1 && variant <= kNumHorizons && kHorizons[variant - 1].structured;
}

size_t variant_lds_bytes(int variant) {
    return (variant >= 1 && variant <= kNumHorizons) ? kHorizons[variant - 1].lds_bytes : 0;
}

hipError_t launch_solve(int variant, int form, const DevCfg& cfg, const double* d_in, int batch, double* d_x, double* d_fm,
                        int* d_status, int* d_iters, double* dbgM, double* dbgL, unsigned long long* stamps,
                        hipStream_t stream) {
    int id = 0;
#define X(N, NS, HC)                                                                                                  \
    if (variant == ++id) {                                                                                            \
        if (stamps != nullptr || dbgM != nullptr || dbgL != nullptr)                                                  \
            return launch_solve_dims<N, NS, HC, true>(form, cfg, d_in, batch, d_x, d_fm, d_status, d_iters, dbgM, dbgL,   \
                                                      stamps, stream);                                                \
        return launch_solve_dims<N, NS, HC, false>(form, cfg, d_in, batch, d_x, d_fm, d_status, d_iters, dbgM, dbgL,      \
                                                   nullptr, stream);                                                  \
    }
#include "vsmpc_horizons.def"
#undef X
    return hipErrorInvalidValue;
}

hipError_t launch_linearize(int variant, const DevCfg& cfg, const double* d_in, int batch, double* A, double* Bj,
                            double* Bt, double* c, hipStream_t stream) {
    int id = 0;
#define X(N, NS, HC) \
    if (variant == ++id) return launch_linearize_dims<N, NS, HC>(cfg, d_in, batch, A, Bj, Bt, c, stream);
#include "vsmpc_horizons.def"
#undef X
    return hipErrorInvalidValue;
}
#endif  // common part

}  // namespace vsmpc
