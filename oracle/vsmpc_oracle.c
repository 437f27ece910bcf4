/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (the reference has no tests or vectors for
 * this path and cannot be built or imported in this image).
 *
 * Plain-C restatement of the reference's multi-rate MPC hot path with the SAME ALGORITHMIC SHAPE as the
 * reference, to be used (a) as a second, independent checker next to oracle/vsmpc_ref.py and (b) as the timed
 * CPU baseline ("port") of bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 *
 *   assembly   dense H, g, A, l, u in the reference's plugin / row order
 *              (momentum-based-linear-mpc-lib/src/IMPCProblem/IMPCProblem.cpp:150-194,
 *               .../variableSamplingMPC/systemDynamicsVSMPC.cpp:79-103,288-319,384-461,
 *               .../constraintsVSMPC.cpp:45-51,61-142,206-247,338-374, .../costsVSMPC.cpp:121-181,369-413,468-487,
 *               558-592, utils/src/JetModel.cpp:10-114)
 *   dense->CSC  the sparseView() of IMPCProblem.cpp:211 (structural pattern is kept so the symbolic analysis is
 *              done once, as OsqpEigen does after initSolver, IMPCProblem.cpp:221-255)
 *   solve      the QP arithmetic of the reference lives in un-vendored third-party code:
 *              osqp-eigen 0.11.0 -> libosqp 1.0.0 -> libqdldl 0.1.8 (pixi.lock:317,232,237), called at
 *              IMPCProblem.cpp:140-145 (warm start on, polish on, defaults otherwise), :263-279, :296.
 *              Restated here from the PUBLISHED algorithm (Stellato, Banjac, Goulart, Bemporad, Boyd, "OSQP: an
 *              operator splitting solver for quadratic programs", Math. Prog. Comp. 2020): Ruiz equilibration
 *              (10 passes) + cost scaling, KKT [[P+sigma I, A'],[A, -diag(1/rho)]] with rho_eq = 1e3 rho,
 *              sparse LDL' (up-looking, elimination tree; T. Davis, "Algorithm 849", the algorithm QDLDL
 *              implements) under a minimum-degree ordering, ADMM with relaxation alpha = 1.6, termination test
 *              every 25 iterations at eps_abs = eps_rel = 1e-3, adaptive rho (tolerance 5), polish with
 *              delta = 1e-6 and 3 refinement steps.  Defaults are from memory of OSQP 1.0 and unverified
 *              (source not in this image).  Deviation: OSQP picks the adaptive-rho interval from setup timing;
 *              a fixed interval of 50 iterations is used here so results are deterministic.
 *   cold start each instance is solved from x = y = 0 (a batch of independent instances has no previous tick).
 *
 * Build: make -C oracle  ->  oracle/_build/liboracle.so
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NX 26
#define NJ 8
#define NT 4

typedef struct vso_config {  /* same layout as vsmpc_config (include/vsmpc.h) */
    int n_iter, n_iter_small, control_horizon, use_jet_dynamic;
    double period_mpc, period_small, period_large;
    double w_com_pos[3], w_com_pos_err[3], w_lin_mom[3], w_rpy[3], w_rpy_err[3], w_ang_mom[3];
    double w_delta_joint[8];
    double w_throttle, w_initial_throttle, w_reg_joint_pos, throttle_min, throttle_max;
} vso_config;

/* input record offsets (include/vsmpc.h) */
enum { IN_X0 = 0, IN_MASS = 26, IN_WRB = 27, IN_OMEGA = 36, IN_ALPHA = 39, IN_GRAV = 40, IN_AMOM = 43, IN_LLIN = 67,
       IN_LANG = 91, IN_INERTIA = 115, IN_RPY = 124, IN_PREF = 127, IN_RPYINIT = 130, IN_T0 = 133, IN_TD0 = 137,
       IN_UPREV = 141, IN_TDES = 145, IN_TDDES = 149, IN_QERR = 153, IN_HOLD = 161, IN_XREF = 162 };

/* ------------------------------------------------------------------ jet model (utils/src/JetModel.cpp:10-114) */
static const double JC[13] = {-4.64730485e-01, -8.13171858e+00, -6.19539230e+00, 6.61113140e-01, 1.67673231e+00,
                              -4.83287064e-01, 8.77996617e+00,  -1.01096376e+00, -5.86442286e-01, 5.19093322e-01,
                              -4.23782666e-01, -1.45705257e+00, -7.83052261e-03};
static const double JN[4] = {108.309, 65.793, 47.333, 31.483};
static double jet_f(double T, double Td) { return JC[0] + JC[1] * T + JC[2] * Td + JC[3] * T * Td + JC[4] * pow(T, 2.0) + JC[5] * pow(Td, 2.0); }
static double jet_g(double T, double Td) { return JC[6] + JC[7] * T + JC[8] * Td + JC[9] * T * Td + JC[10] * pow(T, 2.0) + JC[11] * pow(Td, 2.0); }
static double jet_df_dT(double T, double Td) { return JC[1] + JC[3] * Td + 2 * JC[4] * T; }
static double jet_df_dTd(double T, double Td) { return JC[2] + JC[3] * T + 2 * JC[5] * Td; }
static double jet_dg_dT(double T, double Td) { return JC[7] + JC[9] * Td + 2 * JC[10] * T; }
static double jet_dg_dTd(double T, double Td) { return JC[8] + JC[9] * T + 2 * JC[11] * Td; }
static double jet_v(double u) { return u + JC[12] * pow(u, 2.0); }
static double std_T(double T) { return (T - JN[0]) / JN[1]; }
static double std_Td(double Td) { return Td / JN[1]; }
static double std_U(double u) { return (u - JN[2]) / JN[3]; }
static double v_of_throttle(double u) { return jet_v(std_U(u)); }
/* systemDynamicsVSMPC.cpp:431-461 */
static double dyn_F(double T, double Td) { return jet_f(std_T(T), std_Td(Td)) * JN[1]; }
static double dyn_G(double T, double Td) { return jet_g(std_T(T), std_Td(Td)) * JN[1]; }
static double dyn_dh_dT(double T, double Td, double thr) { return jet_df_dT(std_T(T), std_Td(Td)) + jet_dg_dT(std_T(T), std_Td(Td)) * jet_v(std_U(thr)); }
static double dyn_dh_dTd(double T, double Td, double thr) { return jet_df_dTd(std_T(T), std_Td(Td)) + jet_dg_dTd(std_T(T), std_Td(Td)) * jet_v(std_U(thr)); }

/* ------------------------------------------------------------------ sizes */
static int n_vblocks(const vso_config* c) { return c->control_horizon - c->n_iter_small + 1; }
static int n_var(const vso_config* c) { return NX * (c->n_iter + 1) + NJ * c->control_horizon + NT * n_vblocks(c); }
static int n_con(const vso_config* c) { return NX * (c->n_iter + 1) + NT * (c->n_iter - c->n_iter_small + 1); }
static int n_in(const vso_config* c) { return IN_XREF + 12 * (c->n_iter - c->n_iter_small + 1); }
int vso_sizes(const vso_config* c, int* nvar, int* ncon, int* nin) {
    *nvar = n_var(c); *ncon = n_con(c); *nin = n_in(c);
    return 0;
}

/* ------------------------------------------------------------------ linearisation */
void vso_dt_schedule(const vso_config* c, double* dt) { /* constraintsVSMPC.cpp:45-51,78-84,156-159 */
    const double nS = (double)c->n_iter_small;
    const double beta2 = (c->period_large - nS * c->period_small) / (nS * (nS - 1.0));
    const double beta1 = c->period_small - beta2;
    for (int i = 0; i < c->n_iter; ++i) {
        if (i < c->n_iter_small) {
            const double a = (double)(i + 1), b = (double)i;
            dt[i] = (beta1 * a + beta2 * a * a) - (beta1 * b + beta2 * b * b);
        } else dt[i] = c->period_large;
    }
}

void vso_linearize(const vso_config* c, const double* in, double* A, double* Bj, double* Bt, double* cv) {
    memset(A, 0, sizeof(double) * NX * NX);
    memset(Bj, 0, sizeof(double) * NX * NJ);
    memset(Bt, 0, sizeof(double) * NX * NT);
    memset(cv, 0, sizeof(double) * NX);
    const double m = in[IN_MASS];
    const double* R = in + IN_WRB;
    const double* w = in + IN_OMEGA;
    const double* I = in + IN_INERTIA;
    /* angular: A[rpy,angMom] = W^-1 I^-1 (systemDynamicsVSMPC.cpp:86-87,140-147) */
    const double det = I[0] * (I[4] * I[8] - I[5] * I[7]) - I[1] * (I[3] * I[8] - I[5] * I[6]) + I[2] * (I[3] * I[7] - I[4] * I[6]);
    double Ii[9];
    Ii[0] = (I[4] * I[8] - I[5] * I[7]) / det; Ii[1] = (I[2] * I[7] - I[1] * I[8]) / det; Ii[2] = (I[1] * I[5] - I[2] * I[4]) / det;
    Ii[3] = (I[5] * I[6] - I[3] * I[8]) / det; Ii[4] = (I[0] * I[8] - I[2] * I[6]) / det; Ii[5] = (I[2] * I[3] - I[0] * I[5]) / det;
    Ii[6] = (I[3] * I[7] - I[4] * I[6]) / det; Ii[7] = (I[1] * I[6] - I[0] * I[7]) / det; Ii[8] = (I[0] * I[4] - I[1] * I[3]) / det;
    const double r = in[IN_RPY], p = in[IN_RPY + 1];
    const double Wi[9] = {1.0, sin(r) * tan(p), cos(r) * tan(p), 0.0, cos(r), -sin(r), 0.0, sin(r) / cos(p), cos(r) / cos(p)};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int q = 0; q < 3; ++q) s += Wi[3 * i + q] * Ii[3 * q + j];
            A[(6 + i) * NX + 9 + j] = s;
        }
    const double S[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0}; /* FlightControlUtils.cpp:77-85 */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[(9 + i) * NX + 9 + j] -= S[3 * i + j];                     /* :90-91 */
            A[(3 + i) * NX + 3 + j] -= S[3 * i + j];                     /* :301-302 */
            A[i * NX + 3 + j] = 1.0 / m * R[3 * i + j];                  /* :296-297 */
        }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 4; ++j) {
            A[(9 + i) * NX + 12 + j] = in[IN_AMOM + (3 + i) * 4 + j];    /* :92-93 */
            A[(3 + i) * NX + 12 + j] = in[IN_AMOM + i * 4 + j];          /* :303-304 */
        }
        for (int j = 0; j < NJ; ++j) {
            Bj[(9 + i) * NJ + j] = in[IN_LANG + i * NJ + j];             /* :94-95 */
            Bj[(3 + i) * NJ + j] = in[IN_LLIN + i * NJ + j];             /* :305-306 */
        }
        A[(23 + i) * NX + 6 + i] = 1.0;                                  /* :98-99 */
        cv[23 + i] = -in[IN_RPYINIT + i];                                /* :100 */
        A[(20 + i) * NX + i] = 1.0;                                      /* :314-315 */
        cv[20 + i] = -in[IN_PREF + i];                                   /* :316 */
        cv[3 + i] = in[IN_ALPHA] * m * (R[i] * in[IN_GRAV] + R[3 + i] * in[IN_GRAV + 1] + R[6 + i] * in[IN_GRAV + 2]); /* :307-309 */
    }
    if (c->use_jet_dynamic) {                                            /* :390-421 */
        for (int i = 0; i < NT; ++i) {
            const double T0 = in[IN_T0 + i], Td0 = in[IN_TD0 + i], up = in[IN_UPREV + i];
            const double dhT = dyn_dh_dT(T0, Td0, up), dhTd = dyn_dh_dTd(T0, Td0, up);
            A[(12 + i) * NX + 16 + i] = 1.0;
            A[(16 + i) * NX + 12 + i] = dhT;
            A[(16 + i) * NX + 16 + i] += dhTd;
            Bt[(16 + i) * NT + i] = dyn_G(in[IN_TDES + i], in[IN_TDDES + i]);
            cv[16 + i] = dyn_F(T0, Td0) - dhT * T0 - dhTd * Td0;
        }
    } else {
        for (int i = 0; i < NT; ++i) Bt[(12 + i) * NT + i] = 1.0;        /* :424-425 */
    }
}

/* ------------------------------------------------------------------ dense assembly, reference order */
void vso_assemble_dense(const vso_config* c, const double* in, double* H, double* g, double* Ac, double* lo, double* hi) {
    const int N = c->n_iter, nS = c->n_iter_small, Hc = c->control_horizon, nvb = n_vblocks(c);
    const int nv = n_var(c), nc = n_con(c), offJ = NX * (N + 1), offV = offJ + NJ * Hc;
    double A[NX * NX], Bj[NX * NJ], Bt[NX * NT], cv[NX], dt[64], q[NX];
    vso_linearize(c, in, A, Bj, Bt, cv);
    vso_dt_schedule(c, dt);
    memset(H, 0, sizeof(double) * (size_t)nv * nv);
    memset(g, 0, sizeof(double) * nv);
    memset(Ac, 0, sizeof(double) * (size_t)nc * nv);
    memset(lo, 0, sizeof(double) * nc);
    memset(hi, 0, sizeof(double) * nc);
    memset(q, 0, sizeof(q));
    for (int i = 0; i < 3; ++i) { /* costsVSMPC.cpp:78-93 */
        q[i] = c->w_com_pos[i]; q[3 + i] = c->w_lin_mom[i]; q[6 + i] = c->w_rpy[i];
        q[9 + i] = c->w_ang_mom[i]; q[20 + i] = c->w_com_pos_err[i]; q[23 + i] = c->w_rpy_err[i];
    }
#define HH(r, cc) H[(size_t)(r) * nv + (cc)]
#define AA(r, cc) Ac[(size_t)(r) * nv + (cc)]
    for (int i = 1; i <= N; ++i) { /* ReferenceTrackingCost, costsVSMPC.cpp:166-178,191-200 */
        const int col = (i - 1) < nS ? 0 : (i - 1) - nS;
        for (int r = 0; r < NX; ++r) {
            HH(i * NX + r, i * NX + r) += q[r];
            if (r < 12) g[i * NX + r] += -q[r] * in[IN_XREF + col * 12 + r];
        }
    }
    for (int i = 0; i < Hc; ++i) /* RegualarizationCost, costsVSMPC.cpp:375-381 */
        for (int r = 0; r < NJ; ++r) HH(offJ + i * NJ + r, offJ + i * NJ + r) += c->w_delta_joint[r];
    for (int i = 0; i < Hc - nS; ++i) /* :383-409 */
        for (int r = 0; r < NT; ++r) {
            const int a = offV + i * NT + r, b = offV + (i + 1) * NT + r;
            HH(a, a) += c->w_throttle; HH(b, a) -= c->w_throttle; HH(a, b) -= c->w_throttle; HH(b, b) += c->w_throttle;
        }
    double vprev[NT];
    for (int r = 0; r < NT; ++r) { /* ThrottleInitialValueCost, costsVSMPC.cpp:468-487 */
        vprev[r] = v_of_throttle(in[IN_UPREV + r]);
        HH(offV + r, offV + r) += c->w_initial_throttle;
        g[offV + r] += -c->w_initial_throttle * vprev[r];
    }
    for (int i = 0; i < Hc; ++i) /* JointPositionRegularizationCost, costsVSMPC.cpp:558-592 */
        for (int r = 0; r < NJ; ++r) {
            HH(offJ + i * NJ + r, offJ + i * NJ + r) += c->w_reg_joint_pos;
            g[offJ + i * NJ + r] += c->w_reg_joint_pos * in[IN_QERR + r];
        }
    for (int i = 0; i < N; ++i) { /* ConstraintSystemDynamicVS, constraintsVSMPC.cpp:76-131 */
        const double d = dt[i];
        const int jb = i < Hc ? i : Hc - 1;
        const int tb = i < nS ? 0 : (i < Hc ? i - (nS - 1) : Hc - nS);
        for (int r = 0; r < NX; ++r) {
            for (int cc = 0; cc < NX; ++cc) AA(i * NX + r, i * NX + cc) = (r == cc ? 1.0 : 0.0) + d * A[r * NX + cc];
            AA(i * NX + r, (i + 1) * NX + r) = -1.0;
            for (int cc = 0; cc < NJ; ++cc) AA(i * NX + r, offJ + jb * NJ + cc) = d * Bj[r * NJ + cc];
            for (int cc = 0; cc < NT; ++cc) AA(i * NX + r, offV + tb * NT + cc) = d * Bt[r * NT + cc];
            lo[i * NX + r] = -d * cv[r];
            hi[i * NX + r] = -d * cv[r];
        }
    }
    const int r0 = N * NX; /* ConstraintInitialState, IQPUtilsMPC.cpp:71-92 */
    for (int r = 0; r < NX; ++r) { AA(r0 + r, r) = 1.0; lo[r0 + r] = hi[r0 + r] = in[IN_X0 + r]; }
    const int r1 = r0 + NX; /* ThrottleConstraint, constraintsVSMPC.cpp:338-365 */
    const double vmin = v_of_throttle(c->throttle_min), vmax = v_of_throttle(c->throttle_max);
    for (int i = 0; i < nvb; ++i)
        for (int r = 0; r < NT; ++r) {
            AA(r1 + i * NT + r, offV + i * NT + r) = 1.0;
            if (in[IN_HOLD] != 0.0 && i == 0) { lo[r1 + r] = hi[r1 + r] = vprev[r]; }
            else { lo[r1 + i * NT + r] = vmin; hi[r1 + i * NT + r] = vmax; }
        }
#undef HH
#undef AA
}

/* ================================================================== sparse LDL' (up-looking, elimination tree) */
typedef struct {
    int n;
    int *Lp, *Li, *Parent, *Lnz, *Flag, *Pattern;
    double *Lx, *D, *Y;
} ldl_t;

static void ldl_free(ldl_t* f) {
    free(f->Lp); free(f->Li); free(f->Parent); free(f->Lnz); free(f->Flag); free(f->Pattern); free(f->Lx); free(f->D); free(f->Y);
    memset(f, 0, sizeof(*f));
}

/* symbolic analysis of an upper-triangular CSC matrix (entries with row <= col) */
static void ldl_symbolic(ldl_t* f, int n, const int* Ap, const int* Ai) {
    f->n = n;
    f->Lp = (int*)malloc(sizeof(int) * (n + 1)); f->Parent = (int*)malloc(sizeof(int) * n);
    f->Lnz = (int*)malloc(sizeof(int) * n); f->Flag = (int*)malloc(sizeof(int) * n); f->Pattern = (int*)malloc(sizeof(int) * n);
    f->D = (double*)malloc(sizeof(double) * n); f->Y = (double*)malloc(sizeof(double) * n);
    for (int k = 0; k < n; ++k) {
        f->Parent[k] = -1; f->Flag[k] = k; f->Lnz[k] = 0;
        for (int p = Ap[k]; p < Ap[k + 1]; ++p) {
            int i = Ai[p];
            if (i < k)
                for (; f->Flag[i] != k; i = f->Parent[i]) {
                    if (f->Parent[i] == -1) f->Parent[i] = k;
                    f->Lnz[i]++;
                    f->Flag[i] = k;
                }
        }
    }
    f->Lp[0] = 0;
    for (int k = 0; k < n; ++k) f->Lp[k + 1] = f->Lp[k] + f->Lnz[k];
    f->Li = (int*)malloc(sizeof(int) * (f->Lp[n] > 0 ? f->Lp[n] : 1));
    f->Lx = (double*)malloc(sizeof(double) * (f->Lp[n] > 0 ? f->Lp[n] : 1));
}

static int ldl_numeric(ldl_t* f, const int* Ap, const int* Ai, const double* Ax) {
    const int n = f->n;
    for (int k = 0; k < n; ++k) {
        f->Y[k] = 0.0;
        int top = n;
        f->Flag[k] = k; f->Lnz[k] = 0;
        for (int p = Ap[k]; p < Ap[k + 1]; ++p) {
            int i = Ai[p];
            if (i <= k) {
                f->Y[i] += Ax[p];
                int len = 0;
                for (; f->Flag[i] != k; i = f->Parent[i]) { f->Pattern[len++] = i; f->Flag[i] = k; }
                while (len > 0) f->Pattern[--top] = f->Pattern[--len];
            }
        }
        f->D[k] = f->Y[k];
        f->Y[k] = 0.0;
        for (; top < n; ++top) {
            const int i = f->Pattern[top];
            const double yi = f->Y[i];
            f->Y[i] = 0.0;
            const int p2 = f->Lp[i] + f->Lnz[i];
            int p;
            for (p = f->Lp[i]; p < p2; ++p) f->Y[f->Li[p]] -= f->Lx[p] * yi;
            const double lki = yi / f->D[i];
            f->D[k] -= lki * yi;
            f->Li[p] = k;
            f->Lx[p] = lki;
            f->Lnz[i]++;
        }
        if (f->D[k] == 0.0) return k + 1;
    }
    return 0;
}

static void ldl_solve(const ldl_t* f, double* x) {
    const int n = f->n;
    for (int j = 0; j < n; ++j) { const double xj = x[j]; for (int p = f->Lp[j]; p < f->Lp[j] + f->Lnz[j]; ++p) x[f->Li[p]] -= f->Lx[p] * xj; }
    for (int j = 0; j < n; ++j) x[j] /= f->D[j];
    for (int j = n - 1; j >= 0; --j) { double s = x[j]; for (int p = f->Lp[j]; p < f->Lp[j] + f->Lnz[j]; ++p) s -= f->Lx[p] * x[f->Li[p]]; x[j] = s; }
}

/* minimum-degree ordering of a symmetric pattern given as triplets (r,c), bitset elimination graph */
static void min_degree(int n, int nnz, const int* tr, const int* tc, int* perm) {
    const int W = (n + 63) / 64;
    unsigned long long* adj = (unsigned long long*)calloc((size_t)n * W, sizeof(unsigned long long));
    char* done = (char*)calloc(n, 1);
    for (int e = 0; e < nnz; ++e) {
        const int a = tr[e], b = tc[e];
        if (a == b) continue;
        adj[(size_t)a * W + (b >> 6)] |= 1ull << (b & 63);
        adj[(size_t)b * W + (a >> 6)] |= 1ull << (a & 63);
    }
    int* deg = (int*)malloc(sizeof(int) * n);
    for (int i = 0; i < n; ++i) { int d = 0; for (int w = 0; w < W; ++w) d += __builtin_popcountll(adj[(size_t)i * W + w]); deg[i] = d; }
    for (int k = 0; k < n; ++k) {
        int best = -1;
        for (int i = 0; i < n; ++i) if (!done[i] && (best < 0 || deg[i] < deg[best])) best = i;
        perm[k] = best;
        done[best] = 1;
        unsigned long long* nb = adj + (size_t)best * W;
        for (int w = 0; w < W; ++w) {
            unsigned long long bits = nb[w];
            while (bits) {
                const int u = (w << 6) + __builtin_ctzll(bits);
                bits &= bits - 1;
                unsigned long long* au = adj + (size_t)u * W;
                for (int w2 = 0; w2 < W; ++w2) au[w2] |= nb[w2];
                au[u >> 6] &= ~(1ull << (u & 63));
                au[best >> 6] &= ~(1ull << (best & 63));
                int d = 0;
                for (int w2 = 0; w2 < W; ++w2) d += __builtin_popcountll(au[w2]);
                deg[u] = d;
            }
        }
    }
    free(adj); free(done); free(deg);
}

/* symmetric matrix from triplets (any triangle) -> permuted upper-triangular CSC; map[e] = CSC slot of triplet e */
typedef struct { int n, nnz; int *Ap, *Ai; double* Ax; int* map; } csc_t;
static void csc_free(csc_t* m) { free(m->Ap); free(m->Ai); free(m->Ax); free(m->map); memset(m, 0, sizeof(*m)); }
static void build_permuted_upper(csc_t* out, int n, int nnz, const int* tr, const int* tc, const int* iperm) {
    out->n = n; out->nnz = nnz;
    out->Ap = (int*)calloc(n + 1, sizeof(int)); out->Ai = (int*)malloc(sizeof(int) * nnz);
    out->Ax = (double*)calloc(nnz, sizeof(double)); out->map = (int*)malloc(sizeof(int) * nnz);
    int* rr = (int*)malloc(sizeof(int) * nnz); int* cc = (int*)malloc(sizeof(int) * nnz);
    for (int e = 0; e < nnz; ++e) {
        int a = iperm[tr[e]], b = iperm[tc[e]];
        if (a > b) { int t = a; a = b; b = t; }
        rr[e] = a; cc[e] = b;
        out->Ap[b + 1]++;
    }
    for (int j = 0; j < n; ++j) out->Ap[j + 1] += out->Ap[j];
    int* next = (int*)malloc(sizeof(int) * n);
    memcpy(next, out->Ap, sizeof(int) * n);
    for (int e = 0; e < nnz; ++e) { const int s = next[cc[e]]++; out->Ai[s] = rr[e]; out->map[e] = s; }
    free(rr); free(cc); free(next);
}

/* ================================================================== OSQP-style workspace */
typedef struct {
    vso_config cfg;
    int n, m;
    /* structural pattern of P (upper) and A as triplets (row, col) into the dense arrays */
    int pnz, anz; int *Pr, *Pc, *Ar, *Ac;
    /* KKT: triplets = [P entries | sigma diag (n) | A entries as (col, n+row) | rho diag (m)] */
    int knz; int *Kr, *Kc; int *perm, *iperm; csc_t K; ldl_t F;
    /* scaled problem data */
    double *Px, *Ax, *q, *l, *u, *D, *E, *Dinv, *Einv, cscale, *rho_vec, *rho_inv;
    /* iterates (scaled) */
    double *x, *z, *y, *xprev, *zprev, *xt, *zt, *rhs, *tmpn, *tmpm, *Axv, *Pxv, *Aty;
    /* dense assembly buffers */
    double *Hd, *gd, *Acd, *lod, *hid;
    int last_iters, last_polished, last_rho_updates;
    /* warm start (IMPCProblem.cpp:140 setWarmStart(true)): the previous solve's primal / dual (unscaled) and rho */
    int warm, have_prev;
    double *xw, *yw, rho_prev;
} ws_t;

static void spmv_A(const ws_t* w, const double* x, double* y) { /* y = A x (scaled values) */
    memset(y, 0, sizeof(double) * w->m);
    for (int e = 0; e < w->anz; ++e) y[w->Ar[e]] += w->Ax[e] * x[w->Ac[e]];
}
static void spmv_At(const ws_t* w, const double* yv, double* x) {
    memset(x, 0, sizeof(double) * w->n);
    for (int e = 0; e < w->anz; ++e) x[w->Ac[e]] += w->Ax[e] * yv[w->Ar[e]];
}
static void spmv_P(const ws_t* w, const double* x, double* y) { /* symmetric, upper triangle stored */
    memset(y, 0, sizeof(double) * w->n);
    for (int e = 0; e < w->pnz; ++e) {
        const int r = w->Pr[e], c = w->Pc[e];
        y[r] += w->Px[e] * x[c];
        if (r != c) y[c] += w->Px[e] * x[r];
    }
}
static double norm_inf(const double* v, int n) { double s = 0.0; for (int i = 0; i < n; ++i) { const double a = fabs(v[i]); if (a > s) s = a; } return s; }
static double norm_inf_scaled(const double* v, const double* s, int n) { double r = 0.0; for (int i = 0; i < n; ++i) { const double a = fabs(v[i] * s[i]); if (a > r) r = a; } return r; }

void* vso_workspace_create(const vso_config* c) {
    ws_t* w = (ws_t*)calloc(1, sizeof(ws_t));
    w->cfg = *c;
    const int n = n_var(c), m = n_con(c), nin = n_in(c);
    w->n = n; w->m = m;
    w->Hd = (double*)malloc(sizeof(double) * (size_t)n * n); w->gd = (double*)malloc(sizeof(double) * n);
    w->Acd = (double*)malloc(sizeof(double) * (size_t)m * n); w->lod = (double*)malloc(sizeof(double) * m); w->hid = (double*)malloc(sizeof(double) * m);
    /* structural pattern: assemble with an input record that makes every structural entry non-zero */
    double* probe = (double*)malloc(sizeof(double) * nin);
    for (int i = 0; i < nin; ++i) probe[i] = 0.37 + 0.011 * (double)(i % 17);
    probe[IN_MASS] = 70.0; probe[IN_HOLD] = 0.0;
    const double I0[9] = {8.0, 0.1, 0.2, 0.1, 7.0, 0.3, 0.2, 0.3, 2.0};
    memcpy(probe + IN_INERTIA, I0, sizeof(I0));
    for (int i = 0; i < 4; ++i) { probe[IN_T0 + i] = 160.0 + i; probe[IN_TDES + i] = 161.0 + i; probe[IN_UPREV + i] = 70.0 + i; }
    vso_assemble_dense(c, probe, w->Hd, w->gd, w->Acd, w->lod, w->hid);
    free(probe);
    int pnz = 0, anz = 0;
    for (int r = 0; r < n; ++r) for (int cc = r; cc < n; ++cc) if (w->Hd[(size_t)r * n + cc] != 0.0) pnz++;
    for (int r = 0; r < m; ++r) for (int cc = 0; cc < n; ++cc) if (w->Acd[(size_t)r * n + cc] != 0.0) anz++;
    w->pnz = pnz; w->anz = anz;
    w->Pr = (int*)malloc(sizeof(int) * pnz); w->Pc = (int*)malloc(sizeof(int) * pnz);
    w->Ar = (int*)malloc(sizeof(int) * anz); w->Ac = (int*)malloc(sizeof(int) * anz);
    pnz = anz = 0;
    for (int r = 0; r < n; ++r) for (int cc = r; cc < n; ++cc) if (w->Hd[(size_t)r * n + cc] != 0.0) { w->Pr[pnz] = r; w->Pc[pnz] = cc; pnz++; }
    for (int r = 0; r < m; ++r) for (int cc = 0; cc < n; ++cc) if (w->Acd[(size_t)r * n + cc] != 0.0) { w->Ar[anz] = r; w->Ac[anz] = cc; anz++; }
    /* KKT triplets */
    w->knz = pnz + n + anz + m;
    w->Kr = (int*)malloc(sizeof(int) * w->knz); w->Kc = (int*)malloc(sizeof(int) * w->knz);
    int e = 0;
    for (int i = 0; i < pnz; ++i, ++e) { w->Kr[e] = w->Pr[i]; w->Kc[e] = w->Pc[i]; }
    for (int i = 0; i < n; ++i, ++e) { w->Kr[e] = i; w->Kc[e] = i; }
    for (int i = 0; i < anz; ++i, ++e) { w->Kr[e] = w->Ac[i]; w->Kc[e] = n + w->Ar[i]; }
    for (int i = 0; i < m; ++i, ++e) { w->Kr[e] = n + i; w->Kc[e] = n + i; }
    const int N = n + m;
    w->perm = (int*)malloc(sizeof(int) * N); w->iperm = (int*)malloc(sizeof(int) * N);
    min_degree(N, w->knz, w->Kr, w->Kc, w->perm);
    for (int i = 0; i < N; ++i) w->iperm[w->perm[i]] = i;
    build_permuted_upper(&w->K, N, w->knz, w->Kr, w->Kc, w->iperm);
    ldl_symbolic(&w->F, N, w->K.Ap, w->K.Ai);
#define DV(name, cnt) w->name = (double*)calloc((cnt), sizeof(double))
    DV(Px, pnz); DV(Ax, anz); DV(q, n); DV(l, m); DV(u, m); DV(D, n); DV(E, m); DV(Dinv, n); DV(Einv, m);
    DV(rho_vec, m); DV(rho_inv, m); DV(x, n); DV(z, m); DV(y, m); DV(xprev, n); DV(zprev, m); DV(xt, n); DV(zt, m);
    DV(rhs, N); DV(tmpn, n); DV(tmpm, m); DV(Axv, m); DV(Pxv, n); DV(Aty, n);
#undef DV
    return w;
}

void vso_workspace_free(void* p) {
    ws_t* w = (ws_t*)p;
    if (!w) return;
    free(w->Hd); free(w->gd); free(w->Acd); free(w->lod); free(w->hid);
    free(w->Pr); free(w->Pc); free(w->Ar); free(w->Ac); free(w->Kr); free(w->Kc); free(w->perm); free(w->iperm);
    csc_free(&w->K); ldl_free(&w->F);
    free(w->Px); free(w->Ax); free(w->q); free(w->l); free(w->u); free(w->D); free(w->E); free(w->Dinv); free(w->Einv);
    free(w->rho_vec); free(w->rho_inv); free(w->x); free(w->z); free(w->y); free(w->xprev); free(w->zprev); free(w->xt);
    free(w->zt); free(w->rhs); free(w->tmpn); free(w->tmpm); free(w->Axv); free(w->Pxv); free(w->Aty);
    free(w->xw); free(w->yw);
    free(w);
}

/* OSQP defaults (from memory of libosqp 1.0, unverified) */
#define OSQP_RHO 0.1
#define OSQP_SIGMA 1e-6
#define OSQP_ALPHA 1.6
#define OSQP_EPS_ABS 1e-3
#define OSQP_EPS_REL 1e-3
#define OSQP_MAX_ITER 4000
#define OSQP_SCALING 10
#define OSQP_CHECK_TERMINATION 25
#define OSQP_ADAPTIVE_RHO_INTERVAL 50
#define OSQP_ADAPTIVE_RHO_TOL 5.0
#define OSQP_RHO_EQ_OVER_INEQ 1e3
#define OSQP_RHO_MIN 1e-6
#define OSQP_RHO_MAX 1e6
#define OSQP_POLISH_DELTA 1e-6
#define OSQP_POLISH_REFINE 3
#define MIN_SCALING 1e-4
#define MAX_SCALING 1e4

static double limit_scaling(double v) { if (v < MIN_SCALING) return 1.0; if (v > MAX_SCALING) return MAX_SCALING; return v; }

static void scale_problem(ws_t* w) {
    const int n = w->n, m = w->m;
    for (int i = 0; i < n; ++i) w->D[i] = 1.0;
    for (int i = 0; i < m; ++i) w->E[i] = 1.0;
    w->cscale = 1.0;
    double* dn = w->tmpn; double* en = w->tmpm;
    for (int it = 0; it < OSQP_SCALING; ++it) {
        for (int i = 0; i < n; ++i) dn[i] = 0.0;
        for (int i = 0; i < m; ++i) en[i] = 0.0;
        for (int e = 0; e < w->pnz; ++e) { const double a = fabs(w->Px[e]); const int r = w->Pr[e], c = w->Pc[e]; if (a > dn[c]) dn[c] = a; if (a > dn[r]) dn[r] = a; }
        for (int e = 0; e < w->anz; ++e) { const double a = fabs(w->Ax[e]); if (a > dn[w->Ac[e]]) dn[w->Ac[e]] = a; if (a > en[w->Ar[e]]) en[w->Ar[e]] = a; }
        for (int i = 0; i < n; ++i) dn[i] = 1.0 / sqrt(limit_scaling(dn[i]));
        for (int i = 0; i < m; ++i) en[i] = 1.0 / sqrt(limit_scaling(en[i]));
        for (int e = 0; e < w->pnz; ++e) w->Px[e] *= dn[w->Pr[e]] * dn[w->Pc[e]];
        for (int e = 0; e < w->anz; ++e) w->Ax[e] *= en[w->Ar[e]] * dn[w->Ac[e]];
        for (int i = 0; i < n; ++i) { w->q[i] *= dn[i]; w->D[i] *= dn[i]; }
        for (int i = 0; i < m; ++i) w->E[i] *= en[i];
        /* cost scaling */
        for (int i = 0; i < n; ++i) dn[i] = 0.0;
        for (int e = 0; e < w->pnz; ++e) { const double a = fabs(w->Px[e]); const int r = w->Pr[e], c = w->Pc[e]; if (a > dn[c]) dn[c] = a; if (a > dn[r]) dn[r] = a; }
        double mean = 0.0;
        for (int i = 0; i < n; ++i) mean += dn[i];
        mean /= (double)n;
        double ct = limit_scaling(fmax(mean, norm_inf(w->q, n)));
        ct = 1.0 / ct;
        for (int e = 0; e < w->pnz; ++e) w->Px[e] *= ct;
        for (int i = 0; i < n; ++i) w->q[i] *= ct;
        w->cscale *= ct;
    }
    for (int i = 0; i < n; ++i) w->Dinv[i] = 1.0 / w->D[i];
    for (int i = 0; i < m; ++i) { w->Einv[i] = 1.0 / w->E[i]; w->l[i] *= w->E[i]; w->u[i] *= w->E[i]; }
}

static void set_rho(ws_t* w, double rho) {
    for (int i = 0; i < w->m; ++i) {
        const int eq = fabs(w->u[i] - w->l[i]) < 1e-4;
        w->rho_vec[i] = eq ? OSQP_RHO_EQ_OVER_INEQ * rho : rho;
        w->rho_inv[i] = 1.0 / w->rho_vec[i];
    }
}

static int kkt_refactor(ws_t* w) {
    int e = 0;
    double* Kx = w->K.Ax;
    memset(Kx, 0, sizeof(double) * w->K.nnz);
    for (int i = 0; i < w->pnz; ++i, ++e) Kx[w->K.map[e]] += w->Px[i];
    for (int i = 0; i < w->n; ++i, ++e) Kx[w->K.map[e]] += OSQP_SIGMA;
    for (int i = 0; i < w->anz; ++i, ++e) Kx[w->K.map[e]] += w->Ax[i];
    for (int i = 0; i < w->m; ++i, ++e) Kx[w->K.map[e]] += -w->rho_inv[i];
    return ldl_numeric(&w->F, w->K.Ap, w->K.Ai, Kx);
}

static void kkt_solve(ws_t* w, double* rhs /* length n+m, natural order, overwritten */) {
    const int N = w->n + w->m;
    double* t = (double*)alloca(sizeof(double) * N);
    for (int i = 0; i < N; ++i) t[i] = rhs[w->perm[i]];
    ldl_solve(&w->F, t);
    for (int i = 0; i < N; ++i) rhs[w->perm[i]] = t[i];
}

typedef struct { double prim, dual, prim_norm, dual_norm; } res_t;
static res_t residuals(ws_t* w, const double* x, const double* z, const double* y) {
    res_t r;
    spmv_A(w, x, w->Axv);
    spmv_P(w, x, w->Pxv);
    spmv_At(w, y, w->Aty);
    double pr = 0.0, dr = 0.0;
    for (int i = 0; i < w->m; ++i) { const double a = fabs((w->Axv[i] - z[i]) * w->Einv[i]); if (a > pr) pr = a; }
    for (int i = 0; i < w->n; ++i) { const double a = fabs((w->Pxv[i] + w->q[i] + w->Aty[i]) * w->Dinv[i]); if (a > dr) dr = a; }
    r.prim = pr;
    r.dual = dr / w->cscale;
    r.prim_norm = fmax(norm_inf_scaled(w->Axv, w->Einv, w->m), norm_inf_scaled(z, w->Einv, w->m));
    r.dual_norm = fmax(fmax(norm_inf_scaled(w->Pxv, w->Dinv, w->n), norm_inf_scaled(w->Aty, w->Dinv, w->n)), norm_inf_scaled(w->q, w->Dinv, w->n)) / w->cscale;
    return r;
}

/* polish: reduced KKT on the active constraints guessed from the duals */
static int polish(ws_t* w, res_t admm) {
    const int n = w->n, m = w->m;
    int* act = (int*)malloc(sizeof(int) * m);
    double* bnd = (double*)malloc(sizeof(double) * m);
    int na = 0;
    for (int i = 0; i < m; ++i) {
        if (w->z[i] - w->l[i] < -w->y[i]) { act[na] = i; bnd[na] = w->l[i]; na++; }
        else if (w->u[i] - w->z[i] < w->y[i]) { act[na] = i; bnd[na] = w->u[i]; na++; }
    }
    int* rowmap = (int*)malloc(sizeof(int) * m);
    for (int i = 0; i < m; ++i) rowmap[i] = -1;
    for (int k = 0; k < na; ++k) rowmap[act[k]] = k;
    int ared = 0;
    for (int e = 0; e < w->anz; ++e) if (rowmap[w->Ar[e]] >= 0) ared++;
    const int N = n + na, nnz = w->pnz + n + ared + na;
    int* tr = (int*)malloc(sizeof(int) * nnz); int* tc = (int*)malloc(sizeof(int) * nnz);
    double* tv = (double*)malloc(sizeof(double) * nnz); double* tv0 = (double*)malloc(sizeof(double) * nnz);
    int e = 0;
    for (int i = 0; i < w->pnz; ++i, ++e) { tr[e] = w->Pr[i]; tc[e] = w->Pc[i]; tv[e] = tv0[e] = w->Px[i]; }
    for (int i = 0; i < n; ++i, ++e) { tr[e] = i; tc[e] = i; tv[e] = OSQP_POLISH_DELTA; tv0[e] = 0.0; }
    for (int i = 0; i < w->anz; ++i) if (rowmap[w->Ar[i]] >= 0) { tr[e] = w->Ac[i]; tc[e] = n + rowmap[w->Ar[i]]; tv[e] = tv0[e] = w->Ax[i]; ++e; }
    for (int i = 0; i < na; ++i, ++e) { tr[e] = n + i; tc[e] = n + i; tv[e] = -OSQP_POLISH_DELTA; tv0[e] = 0.0; }
    int* perm = (int*)malloc(sizeof(int) * N); int* iperm = (int*)malloc(sizeof(int) * N);
    min_degree(N, nnz, tr, tc, perm);
    for (int i = 0; i < N; ++i) iperm[perm[i]] = i;
    csc_t K; ldl_t F;
    build_permuted_upper(&K, N, nnz, tr, tc, iperm);
    for (int i = 0; i < nnz; ++i) K.Ax[K.map[i]] += tv[i];
    ldl_symbolic(&F, N, K.Ap, K.Ai);
    int ok = ldl_numeric(&F, K.Ap, K.Ai, K.Ax) == 0;
    double* rhs = (double*)malloc(sizeof(double) * N); double* sol = (double*)calloc(N, sizeof(double));
    double* resid = (double*)malloc(sizeof(double) * N); double* t = (double*)malloc(sizeof(double) * N);
    if (ok) {
        for (int i = 0; i < n; ++i) rhs[i] = -w->q[i];
        for (int k = 0; k < na; ++k) rhs[n + k] = bnd[k];
        /* iterative refinement against the unregularised KKT: sol += Khat^-1 (rhs - K sol) */
        for (int it = 0; it <= OSQP_POLISH_REFINE; ++it) {
            memcpy(resid, rhs, sizeof(double) * N);
            for (int i = 0; i < nnz; ++i) {
                const int r = tr[i], c = tc[i];
                resid[r] -= tv0[i] * sol[c];
                if (r != c) resid[c] -= tv0[i] * sol[r];
            }
            for (int i = 0; i < N; ++i) t[i] = resid[perm[i]];
            ldl_solve(&F, t);
            for (int i = 0; i < N; ++i) sol[perm[i]] += t[i];
        }
        /* candidate: x_pol, y_pol (zero on inactive rows), z_pol = A x_pol */
        double* xp = w->xt; double* yp = w->tmpm; double* zp = w->zt;
        memcpy(xp, sol, sizeof(double) * n);
        memset(yp, 0, sizeof(double) * m);
        for (int k = 0; k < na; ++k) yp[act[k]] = sol[n + k];
        spmv_A(w, xp, zp);
        for (int i = 0; i < m; ++i) zp[i] = fmin(fmax(zp[i], w->l[i]), w->u[i]);
        res_t pr = residuals(w, xp, zp, yp);
        /* OSQP accepts the polished point if it improves both residuals (or is tiny) */
        const int better = (pr.prim < admm.prim && pr.dual < admm.dual) || (pr.prim < admm.prim && admm.dual < 1e-10) ||
                           (pr.dual < admm.dual && admm.prim < 1e-10);
        if (better) { memcpy(w->x, xp, sizeof(double) * n); memcpy(w->y, yp, sizeof(double) * m); memcpy(w->z, zp, sizeof(double) * m); }
        else ok = 0;
    }
    free(act); free(bnd); free(rowmap); free(tr); free(tc); free(tv); free(tv0); free(perm); free(iperm);
    csc_free(&K); ldl_free(&F); free(rhs); free(sol); free(resid); free(t);
    return ok;
}

/* Returns 1 = solved, 2 = max iterations, 3 = numerical.  x[nvar], y[ncon] in the reference's units. */
int vso_solve(void* wsp, const double* in, double* x, double* y, int* iters, int* polished, double* res2) {
    ws_t* w = (ws_t*)wsp;
    const int n = w->n, m = w->m;
    /* update(): dense plugin-order assembly (IMPCProblem.cpp:150-194) */
    vso_assemble_dense(&w->cfg, in, w->Hd, w->gd, w->Acd, w->lod, w->hid);
    /* solve(): dense -> sparse (IMPCProblem.cpp:211), data update (:263-277) */
    for (int e = 0; e < w->pnz; ++e) w->Px[e] = w->Hd[(size_t)w->Pr[e] * n + w->Pc[e]];
    for (int e = 0; e < w->anz; ++e) w->Ax[e] = w->Acd[(size_t)w->Ar[e] * n + w->Ac[e]];
    memcpy(w->q, w->gd, sizeof(double) * n);
    memcpy(w->l, w->lod, sizeof(double) * m);
    memcpy(w->u, w->hid, sizeof(double) * m);
    scale_problem(w);
    const int warm = w->warm && w->have_prev;
    double rho = warm ? w->rho_prev : OSQP_RHO;   /* OSQP keeps the adapted rho between solves */
    set_rho(w, rho);
    if (kkt_refactor(w) != 0) return 3;
    memset(w->x, 0, sizeof(double) * n); memset(w->z, 0, sizeof(double) * m); memset(w->y, 0, sizeof(double) * m);
    if (warm) {   /* osqp_warm_start: x, y of the previous solve in the new scaling, z = A x */
        for (int i = 0; i < n; ++i) w->x[i] = w->Dinv[i] * w->xw[i];
        for (int i = 0; i < m; ++i) w->y[i] = w->cscale * w->Einv[i] * w->yw[i];
        spmv_A(w, w->x, w->z);
    }
    int status = 2, it = 0, rho_updates = 0;
    res_t r = {0, 0, 0, 0};
    for (it = 1; it <= OSQP_MAX_ITER; ++it) {
        memcpy(w->xprev, w->x, sizeof(double) * n);
        memcpy(w->zprev, w->z, sizeof(double) * m);
        for (int i = 0; i < n; ++i) w->rhs[i] = OSQP_SIGMA * w->xprev[i] - w->q[i];
        for (int i = 0; i < m; ++i) w->rhs[n + i] = w->zprev[i] - w->rho_inv[i] * w->y[i];
        kkt_solve(w, w->rhs);
        for (int i = 0; i < m; ++i) w->zt[i] = w->zprev[i] + w->rho_inv[i] * (w->rhs[n + i] - w->y[i]);
        for (int i = 0; i < n; ++i) w->x[i] = OSQP_ALPHA * w->rhs[i] + (1.0 - OSQP_ALPHA) * w->xprev[i];
        for (int i = 0; i < m; ++i) {
            const double zr = OSQP_ALPHA * w->zt[i] + (1.0 - OSQP_ALPHA) * w->zprev[i];
            const double zn = fmin(fmax(zr + w->rho_inv[i] * w->y[i], w->l[i]), w->u[i]);
            w->y[i] += w->rho_vec[i] * (zr - zn);
            w->z[i] = zn;
        }
        const int check = (it % OSQP_CHECK_TERMINATION) == 0;
        const int adapt = (it % OSQP_ADAPTIVE_RHO_INTERVAL) == 0;
        if (check || adapt) {
            r = residuals(w, w->x, w->z, w->y);
            const double eps_p = OSQP_EPS_ABS + OSQP_EPS_REL * r.prim_norm;
            const double eps_d = OSQP_EPS_ABS + OSQP_EPS_REL * r.dual_norm;
            if (check && r.prim <= eps_p && r.dual <= eps_d) { status = 1; break; }
            if (adapt) {
                double rn = rho * sqrt((r.prim / (r.prim_norm + 1e-10)) / (r.dual / (r.dual_norm + 1e-10)));
                rn = fmin(fmax(rn, OSQP_RHO_MIN), OSQP_RHO_MAX);
                if (rn > rho * OSQP_ADAPTIVE_RHO_TOL || rn < rho / OSQP_ADAPTIVE_RHO_TOL) {
                    /* y is kept; OSQP rescales nothing else on a rho update */
                    rho = rn;
                    set_rho(w, rho);
                    if (kkt_refactor(w) != 0) return 3;
                    rho_updates++;
                }
            }
        }
    }
    if (it > OSQP_MAX_ITER) it = OSQP_MAX_ITER;
    r = residuals(w, w->x, w->z, w->y);
    const int pol = polish(w, r);
    if (pol) r = residuals(w, w->x, w->z, w->y);
    for (int i = 0; i < n; ++i) x[i] = w->D[i] * w->x[i];
    if (y) for (int i = 0; i < m; ++i) y[i] = w->E[i] * w->y[i] / w->cscale;
    if (w->warm) {
        for (int i = 0; i < n; ++i) w->xw[i] = w->D[i] * w->x[i];
        for (int i = 0; i < m; ++i) w->yw[i] = w->E[i] * w->y[i] / w->cscale;
        w->rho_prev = rho;
        w->have_prev = 1;
    }
    if (iters) *iters = it;
    if (polished) *polished = pol;
    if (res2) { res2[0] = r.prim; res2[1] = r.dual; }
    w->last_iters = it; w->last_polished = pol; w->last_rho_updates = rho_updates;
    return status;
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

int vso_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Times update()+solve() on `batch` instances with `threads` workers (one workspace each, set-up excluded, as the
 * reference's initSolver happens once).  Stops after `budget_s`.  Returns elapsed seconds; *done = instances solved,
 * x_out (may be NULL) receives the primal of every solved instance, stats[0..2] = mean iterations, polished
 * fraction, solved fraction. */
void vso_set_warm_start(void* wsp, int on) {
    ws_t* w = (ws_t*)wsp;
    w->warm = on ? 1 : 0;
    w->have_prev = 0;
    if (on && w->xw == NULL) { w->xw = (double*)calloc(w->n, sizeof(double)); w->yw = (double*)calloc(w->m, sizeof(double)); }
}

double vso_time_batch2(const vso_config* c, const double* in, int batch, int threads, double budget_s, int* done,
                       double* x_out, double* stats, int warm);
double vso_time_batch(const vso_config* c, const double* in, int batch, int threads, double budget_s, int* done,
                      double* x_out, double* stats) {
    return vso_time_batch2(c, in, batch, threads, budget_s, done, x_out, stats, 0);
}

/* warm != 0: every worker warm-starts an instance from the solution (x, y, rho) of the instance it solved before, as the
 * reference does from tick to tick (IMPCProblem.cpp:140); the instances of a batch are neighbouring draws, not consecutive
 * ticks, so this is an upper bound on what a closed loop gains. */
double vso_time_batch2(const vso_config* c, const double* in, int batch, int threads, double budget_s, int* done,
                       double* x_out, double* stats, int warm) {
    const int nv = n_var(c), nin = n_in(c);
    if (threads < 1) threads = 1;
    void** ws = (void**)malloc(sizeof(void*) * threads);
    for (int t = 0; t < threads; ++t) { ws[t] = vso_workspace_create(c); if (warm) vso_set_warm_start(ws[t], 1); }
    int ndone = 0;
    long sum_it = 0; int sum_pol = 0, sum_ok = 0;
    const double t0 = now_s();
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1) reduction(+ : ndone, sum_it, sum_pol, sum_ok)
    for (int b = 0; b < batch; ++b) {
        if (now_s() - t0 > budget_s) continue;
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        double* xb = (double*)malloc(sizeof(double) * nv);
        int it = 0, pol = 0;
        const int st = vso_solve(ws[tid], in + (size_t)b * nin, xb, NULL, &it, &pol, NULL);
        if (x_out) memcpy(x_out + (size_t)b * nv, xb, sizeof(double) * nv);
        free(xb);
        ndone += 1; sum_it += it; sum_pol += pol; sum_ok += (st == 1);
    }
    const double el = now_s() - t0;
    for (int t = 0; t < threads; ++t) vso_workspace_free(ws[t]);
    free(ws);
    *done = ndone;
    if (stats && ndone > 0) { stats[0] = (double)sum_it / ndone; stats[1] = (double)sum_pol / ndone; stats[2] = (double)sum_ok / ndone; }
    return el;
}
