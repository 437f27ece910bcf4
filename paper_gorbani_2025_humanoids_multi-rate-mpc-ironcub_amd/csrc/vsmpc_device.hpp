// Shared device/host definitions for the batched multi-rate MPC kernels (gfx950 only).
//
// Restates, for the device, the pure-scalar parts of the reference path:
//   JetModel            utils/src/JetModel.cpp:10-114
//   JetDynamicVS        momentum-based-linear-mpc-lib/src/variableSamplingMPC/systemDynamicsVSMPC.cpp:431-461
//   move-blocking maps  momentum-based-linear-mpc-lib/src/variableSamplingMPC/constraintsVSMPC.cpp:89-128
// (paths relative to /root/reference/src/flight-controller/).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/vsmpc.h"

#define VS_HD __host__ __device__ __forceinline__
#define VS_DEV __device__ __forceinline__

namespace vsmpc {

constexpr int NX = VSMPC_N_STATES;   // 26
constexpr int NJ = VSMPC_N_JOINTS;   // 8
constexpr int NTH = VSMPC_N_THRUSTS; // 4
// Joint reduction (kernel v24).  Every joint block enters the dynamics only through [Lambda_lin; Lambda_ang] (6 x 8, the
// same matrix for all blocks: the model is LTI over the horizon, constraintsVSMPC.cpp:85-103), so after the orthogonal
// change of variables  W^(1/2) U_i = Q y_i + N n  (thin QR of (Lambda W^(-1/2))^T, W = the joint weights of
// costsVSMPC.cpp:375-381,564-591) the 2-dimensional n sees only |n|^2 / 2 + (N^T b)^T n -- closed form, the same for
// every block -- and the condensed problem keeps NJC = 6 unknowns y_i per block with input matrix R^T, unit weights and
// gradient Q^T b (p0_joint_reduction in vsmpc_kernels.hip; executable model: tests/algo_model.py joint_reduction).
constexpr int NJC = 6;
constexpr int NWROWS = 18;           // weighted state rows: 0..11 and 20..25 (costsVSMPC.cpp:78-93)
constexpr int MAX_STAGES = 40;

// Compile-time problem dimensions (variableSamplingMPC.cpp:42-45).
template <int N_, int NS_, int HC_>
struct Dims {
    static constexpr int N = N_;       // nIter
    static constexpr int NS = NS_;     // nIterSmall
    static constexpr int HC = HC_;     // controlHorizon
    static constexpr int NVB = HC - NS + 1;  // throttle blocks
    static constexpr int NUO = NJ * HC;      // joint increments (the reference's variables: outputs)
    static constexpr int NUY = NJC * HC;     // reduced joint unknowns y (see NJC)
    static constexpr int NU = ((NUY + 15) / 16) * 16;   // joint rows of the condensed problem: y, then dummy unknowns (unit
                                             // diagonal, no coupling) up to the tile boundary the throttle rows start at
    static constexpr int NV = NTH * NVB;     // warped throttles
    static constexpr int NZ = NU + NV;       // condensed unknowns
    static constexpr int NP = ((NZ + 1 + 15) / 16) * 16;  // + gradient row, padded to tiles of 16
    static constexpr int NT = NP / 16;
    static constexpr int NTRI = NT * (NT + 1) / 2;
    static constexpr int NREF = N - NS + 1;
    static constexpr int NIN = VSMPC_IN_XREF + 12 * NREF;
    static constexpr int NXS = NX * (N + 1);
    static constexpr int NVAR = NXS + NUO + NV;
    static constexpr int NCON = NXS + NTH * (N - NS + 1);
    // Y row stride (doubles), congruent 16 mod 32: the four row groups of a b64 operand read hit distinct banks
    static constexpr int YS = ((NP + 16) % 32 == 16) ? NP + 16 : NP + 32;
    static constexpr int TS = 16 * 17;       // tile stride (16 rows, padded row stride 17)
    // workgroup shape: four wavefronts (one per SIMD) for every horizon.  The sensitivity recursion runs one thread per
    // (half, column) pair -- half = linear / angular part -- on 128 columns at a time: CPT columns per thread.
    // Short horizons (NP <= 128) need <= 256 registers and <= 80 KB of LDS, so two workgroups share a CU; long horizons
    // hold 30 accumulator tiles per wavefront in the 512-register budget of one wavefront per SIMD.
    static constexpr int BLOCK = 256;
    static constexpr int NWAVES = BLOCK / 64;
    static constexpr int PCOLS = BLOCK / 2;
    static constexpr int CPT = (NP + PCOLS - 1) / PCOLS;
    static constexpr int WG_PER_CU = NP <= 128 ? 2 : 1;
    // shared panels: a wavefront carries the diagonal tile in lanes 0..15 of its first row slot and PANEL_RPW rows below
    // it in the remaining lanes of PANEL_SLOTS slots of 64 rows; at most NWAVES - 1 wavefronts share a panel
    static constexpr int PANEL_SLOTS = (NP - 16 + 48 * (NWAVES - 1) - 1) / (48 * (NWAVES - 1)) <= 1 ? 1 : 2;
    static_assert((NP - 16 + (64 * PANEL_SLOTS - 16) - 1) / (64 * PANEL_SLOTS - 16) <= NWAVES - 1, "panel 0 leaves one wavefront for the side work");
    // Where the factor lives (v10).  The trailing matrix and, after its column has been the panel, every finished
    // off-diagonal tile of L stay in the REGISTERS of the wavefront that owns the tile.  LDS only ever holds
    //   * a ring of two panel columns (column p while it is factored / applied, column p+1 being handed over),
    //   * the throttle corner (tile rows and columns >= PVT: the box QP works on it), and
    //   * the inverses X_p of the joint diagonal tiles (back-substitution without a chain).
    // That is 13 + 3 + 6 tiles at the paper horizon instead of the 28 + 6 + 13 of an LDS-resident factor, which is
    // what lets two workgroups share one CU (<= 80 KB each).  Nothing lives in global memory for any horizon.
    static constexpr int PVT = NU >> 4;                      // first tile row that contains a throttle row
    static constexpr int RING_A = NT;                        // tiles of an even panel column (at most NT)
    static constexpr int RING_TILES = 2 * NT - 1;            // + an odd one (at most NT - 1)
    static constexpr int NCORNER = (NT - PVT) * (NT - PVT + 1) / 2;
    // the corner sits behind the ring AND behind the scratch of P4..P6 that reuses the (by then dead) ring: the box-QP
    // arrays depend on NV only, so with few joint tile rows they can be longer than the ring itself
    static constexpr int SCRATCH_QP = 3 * NV * (NV + 1) + 2 * TS;
    static constexpr int SCRATCH_P6 = NV * (NV + 1) + 2 * NV * (NV + 1) + NWAVES * NP + NX * (N + 1) + NX * N;
    static constexpr int SCRATCH_TILES = ((SCRATCH_QP > SCRATCH_P6 ? SCRATCH_QP : SCRATCH_P6) + TS - 1) / TS;
    static constexpr int CORNER_TILE0 = RING_TILES > SCRATCH_TILES ? RING_TILES : SCRATCH_TILES;
    static constexpr int L_TILES = CORNER_TILE0 + NCORNER;   // LDS tiles addressed through tile_off()
    // Structured condensing (kernel v13, P1s in vsmpc_kernels.hip): the condensed Hessian from forward / adjoint
    // recursions of 3 HC generator columns + NV throttle columns + the affine column per half, whose forward
    // trajectories (9 N doubles) stay in the registers of the lane that owns the column.  Needs the trajectory to fit
    // the register file and the joint rows to be tile aligned; other horizons condense with the SYRK (P1).
    // Long horizons (one workgroup per CU, 512 registers per lane): the generator columns beyond the 64 lanes of a
    // wavefront (NGX of them per half) ride in the free lanes of the throttle wavefront of their half, the thrust
    // contraction A_mom^T W is summed over the halves by LDS atomics inside the chains (no W array, no contraction
    // phase), and the rows of that array keep only the stages some throttle column at or left of them can see.
    static constexpr int NGEN = 3 * HC;
    static constexpr int NGX = NGEN > 64 ? NGEN - 64 : 0;
    static constexpr bool STRUCT_LONG = N > 18;
    static constexpr bool STRUCT_P1 = NU % 16 == 0 && NV + 1 + NGX <= 64 &&
                                      (STRUCT_LONG ? (NP > 128 && N <= 36) : (NGX == 0 && NP <= 128));
    static constexpr int NJPAIR = HC * (HC + 1) / 2;         // joint block pairs (row block >= column block)
    static_assert(PVT >= 1 && PVT < NT, "throttle corner");
    static_assert(N <= MAX_STAGES, "horizon too long");
    static_assert(NV <= 64, "throttle block must fit one wavefront");
};

// Kernel-argument block: everything that depends only on the configuration.
struct DevCfg {
    double dt[MAX_STAGES];     // per-stage step (constraintsVSMPC.cpp:45-51,78-84)
    double sq[NWROWS];         // sqrt of the 18 non-zero state weights
    double wj[NJ];             // weightDeltaJoint + weightRegularizationJointPos (costsVSMPC.cpp:375-381,564-571)
    double w_reg;              // weightRegularizationJointPos
    double w_thr;              // weightThrottle
    double w_init;             // weightInitialThrottle
    double vmin, vmax;         // constraintsVSMPC.cpp:329-332
    int use_jet;               // useJetDynamic
    int max_as_iter;           // active-set iteration cap
};

// ---------------------------------------------------------------- jet model (JetModel.cpp)
struct Jet {
    static constexpr double c0 = -4.64730485e-01, c1 = -8.13171858e+00, c2 = -6.19539230e+00,
                            c3 = 6.61113140e-01, c4 = 1.67673231e+00, c5 = -4.83287064e-01,
                            c6 = 8.77996617e+00, c7 = -1.01096376e+00, c8 = -5.86442286e-01,
                            c9 = 5.19093322e-01, c10 = -4.23782666e-01, c11 = -1.45705257e+00,
                            c12 = -7.83052261e-03;
    static constexpr double muT = 108.309, sgT = 65.793, muU = 47.333, sgU = 31.483;

    static VS_HD double f(double T, double Td) { return c0 + c1 * T + c2 * Td + c3 * T * Td + c4 * T * T + c5 * Td * Td; }
    static VS_HD double g(double T, double Td) { return c6 + c7 * T + c8 * Td + c9 * T * Td + c10 * T * T + c11 * Td * Td; }
    static VS_HD double df_dT(double T, double Td) { return c1 + c3 * Td + 2 * c4 * T; }
    static VS_HD double df_dTd(double T, double Td) { return c2 + c3 * T + 2 * c5 * Td; }
    static VS_HD double dg_dT(double T, double Td) { return c7 + c9 * Td + 2 * c10 * T; }
    static VS_HD double dg_dTd(double T, double Td) { return c8 + c9 * T + 2 * c11 * Td; }
    static VS_HD double v(double u) { return u + c12 * u * u; }
    // reciprocals as constants: an FP64 division costs ~40 instructions on the device
    static constexpr double isgT = 1.0 / sgT, isgU = 1.0 / sgU;
    static VS_HD double stdT(double T) { return (T - muT) * isgT; }
    static VS_HD double stdTd(double Td) { return Td * isgT; }
    static VS_HD double stdU(double u) { return (u - muU) * isgU; }
    static VS_HD double v_of_throttle(double u_percent) { return v(stdU(u_percent)); }
    // values that are compared with each other exactly (v_min, v_max, the pinned previous throttle) use the reference's
    // division (JetModel.cpp:80-83) so that they round identically to it
    static VS_HD double v_of_throttle_div(double u_percent) { return v((u_percent - muU) / sgU); }
    // JetModel::destandardizeThrottle_u2T (JetModel.cpp:93-109)
    static VS_HD double throttle_of_v(double vv) {
        double u = (-1.0 + sqrt(1.0 + 4.0 * c12 * vv)) / (2.0 * c12);
        u = u * sgU + muU;
        return u < 0.0 ? 0.0 : (u > 100.0 ? 100.0 : u);
    }
    // JetDynamicVS::computeF/computeG/compute_dh_dT/compute_dh_dTDot (systemDynamicsVSMPC.cpp:431-461)
    static VS_HD double F(double T, double Td) { return f(stdT(T), stdTd(Td)) * sgT; }
    static VS_HD double G(double T, double Td) { return g(stdT(T), stdTd(Td)) * sgT; }
    static VS_HD double dh_dT(double T, double Td, double thr) {
        const double a = stdT(T), b = stdTd(Td);
        return df_dT(a, b) + dg_dT(a, b) * v(stdU(thr));
    }
    static VS_HD double dh_dTd(double T, double Td, double thr) {
        const double a = stdT(T), b = stdTd(Td);
        return df_dTd(a, b) + dg_dTd(a, b) * v(stdU(thr));
    }
};

// move blocking (constraintsVSMPC.cpp:89-128)
template <class D>
VS_HD constexpr int joint_block_of_stage(int k) { return k < D::HC ? k : D::HC - 1; }
template <class D>
VS_HD constexpr int throttle_block_of_stage(int k) {
    return k < D::NS ? 0 : (k < D::HC ? k - (D::NS - 1) : D::HC - D::NS);
}

// Internal condensed column order: [y_0..y_{HC-1} | dummies | v_1..v_{NVB-1} | v_0 | gradient | pad].
// v_0 goes last so that the 20-tick throttle hold (constraintsVSMPC.cpp:351) pins the trailing block.
template <class D>
VS_HD constexpr int v_block_of_internal(int q) {  // q in [0, NV): internal throttle index -> reference block
    const int b = q >> 2;
    return b < D::NVB - 1 ? b + 1 : 0;
}
// first stage at which internal column c becomes non-zero in the sensitivity recursion
template <class D>
VS_HD constexpr int col_first_stage(int c) {
    if (c < D::NUY) return c / NJC;
    if (c < D::NU) return 1 << 20;   // dummy unknowns: never
    if (c < D::NZ) {
        const int b = v_block_of_internal<D>(c - D::NU);
        return b == 0 ? 0 : D::NS + b - 1;
    }
    if (c == D::NZ) return 0;
    return 1 << 20;
}
template <class D>
VS_HD constexpr int tile_first_stage(int t) {
    int s = 1 << 20;
    for (int c = 16 * t; c < 16 * t + 16; ++c) {
        const int f = col_first_stage<D>(c);
        s = f < s ? f : s;
    }
    return s;
}

VS_HD constexpr int wrow(int r) { return r < 12 ? r : r + 8; }  // weighted-row index -> state row

// Compile-time table of the lower-triangular 16x16 tiles: (row tile, column tile, first stage at which
// the tile of C = sum_k Y_k^T Y_k becomes non-zero), SORTED by that stage and padded to a multiple of 4
// with never-active dummies.  Entry s is owned by wavefront s % NWAVES, slot s / NWAVES, so at every stage the
// active slots of a wavefront form a prefix and the four wavefronts carry the same number of them (+-1).
//
// PIPE (kernel v27, the structured form's pipelined P3): wavefront 0 is the PANEL wavefront -- it owns no tile; its
// slots list the tiles of tile column 0, whose entries it forms and hands to LDS as the first panel -- and ALL tiles are
// dealt to wavefronts 1..3 in column-major order (the tiles of a column, and therefore the updates of every panel step,
// spread evenly over the three).  A wavefront >= 1 never forms the entries of a column-0 tile: they reach its registers
// factored, when panel 0 is done.  Same indexing (entry q * NWAVES + W), TPW = max(ceil(NTRI / 3), NT) slots.
template <class D, bool PIPE = false>
struct TileTab {
    static constexpr int NOWN = PIPE ? D::NWAVES - 1 : D::NWAVES;
    static constexpr int TPW = PIPE ? ((D::NTRI + NOWN - 1) / NOWN > D::NT ? (D::NTRI + NOWN - 1) / NOWN : D::NT)
                                    : (D::NTRI + D::NWAVES - 1) / D::NWAVES;
    static constexpr int NPAD = TPW * D::NWAVES;
    static constexpr int NEVER = 1 << 20;
    int ti[NPAD];
    int tj[NPAD];
    int ts[NPAD];
    bool valid[NPAD];
    constexpr bool ok(int t) const { return valid[t]; }
    // does wavefront w form the entries of its slot t (P1 / P2)?
    constexpr bool forms(int t, int w) const { return ok(t) && !(PIPE && w > 0 && tj[t] == 0); }
    // does wavefront w hold tile t from P3 on (updates, the finished factor, P5)?
    constexpr bool holds(int t, int w) const { return ok(t) && !(PIPE && w == 0); }
    constexpr TileTab() : ti{}, tj{}, ts{}, valid{} {
        for (int t = 0; t < NPAD; ++t) { ti[t] = 0; tj[t] = 0; ts[t] = NEVER; valid[t] = !PIPE && t < D::NTRI; }
        if (PIPE) {
            for (int i = 0; i < D::NT; ++i) {
                ti[i * D::NWAVES] = i;
                ts[i * D::NWAVES] = tile_first_stage<D>(i);
                valid[i * D::NWAVES] = true;
            }
            int k = 0;
            for (int j = 0; j < D::NT; ++j)
                for (int i = j; i < D::NT; ++i, ++k) {
                    const int t = (k / NOWN) * D::NWAVES + 1 + k % NOWN;
                    ti[t] = i;
                    tj[t] = j;
                    const int a = tile_first_stage<D>(i), b = tile_first_stage<D>(j);
                    ts[t] = a > b ? a : b;
                    valid[t] = true;
                }
            return;
        }
        int t = 0;
        for (int i = 0; i < D::NT; ++i)
            for (int j = 0; j <= i; ++j, ++t) {
                ti[t] = i;
                tj[t] = j;
                const int a = tile_first_stage<D>(i), b = tile_first_stage<D>(j);
                ts[t] = a > b ? a : b;
            }
        for (int a = 1; a < D::NTRI; ++a) {  // stable insertion sort by first stage
            const int ki = ti[a], kj = tj[a], ks = ts[a];
            int b = a - 1;
            while (b >= 0 && ts[b] > ks) { ti[b + 1] = ti[b]; tj[b + 1] = tj[b]; ts[b + 1] = ts[b]; --b; }
            ti[b + 1] = ki; tj[b + 1] = kj; ts[b + 1] = ks;
        }
    }
};

// Number of accumulator slots of wavefront w that are active in SYRK pass m (nodes 2m, 2m+1): the table is sorted by
// first stage, so the active slots of a wavefront are a prefix.  One scalar load per pass instead of 3 registers per slot.
template <class D>
struct NactTab {
    static constexpr int NPASS = (D::N + 1) / 2;
    static constexpr int TPW = (D::NTRI + D::NWAVES - 1) / D::NWAVES;
    int n[NPASS][D::NWAVES];
    constexpr NactTab() : n{} {
        constexpr TileTab<D> tab{};
        for (int m = 0; m < NPASS; ++m) {
            const int last = (2 * m + 1 < D::N) ? 2 * m + 1 : 2 * m;
            for (int w = 0; w < D::NWAVES; ++w) {
                int c = 0;
                for (int q = 0; q < TPW; ++q) c += (last >= tab.ts[q * D::NWAVES + w]) ? 1 : 0;
                n[m][w] = c;
            }
        }
    }
};

// Tile coordinates of a wavefront's slots, two slots per 32-bit word (ti | tj << 8 per slot): a handful of scalar
// registers instead of two address registers per slot.
template <class D>
struct TilePack {
    static constexpr int TPW = (D::NTRI + D::NWAVES - 1) / D::NWAVES;
    static constexpr int NWORDS = (TPW + 1) / 2;
    unsigned w[D::NWAVES][NWORDS];
    constexpr TilePack() : w{} {
        constexpr TileTab<D> tab{};
        for (int wv = 0; wv < D::NWAVES; ++wv)
            for (int q = 0; q < TPW; ++q) {
                const int t = q * D::NWAVES + wv;
                w[wv][q >> 1] |= unsigned(tab.ti[t] | (tab.tj[t] << 8)) << (16 * (q & 1));
            }
    }
};

// configuration scalars kept in LDS (copied out of the kernel arguments once; no SGPRs held for them)
constexpr int CFG_SQ = 0;        // 18 sqrt weights
constexpr int CFG_WJ = 18;       // 8 joint weights
constexpr int CFG_WREG = 26;
constexpr int CFG_WTHR = 27;
constexpr int CFG_WINIT = 28;
constexpr int CFG_VMIN = 29;
constexpr int CFG_VMAX = 30;
constexpr int CFG_SIZE = 32;

}  // namespace vsmpc
