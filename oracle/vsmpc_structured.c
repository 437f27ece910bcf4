/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (timed as bench.py's second CPU line, `cpu_structured`; never shipped, never on the
 * product path).  PARITY UNPINNED like everything under oracle/ (SURVEY.md 8c).
 *
 * The STRUCTURE-EXPLOITING exact solve the HIP kernel runs (csrc/vsmpc_kernels.hip; executable model tests/algo_model.py),
 * as straightforward scalar C on one host thread per instance, so that bench.py can separate the algorithmic gain from
 * the hardware gain: the same inputs, the same algorithm, CPU cores instead of the GPU.
 *
 *   linearise   vso_linearize (vsmpc_oracle.c; systemDynamicsVSMPC.cpp:79-103,288-319,384-429)
 *   reduce      every joint block enters the dynamics through [Lambda_lin; Lambda_ang] (6 x 8, the same for all blocks,
 *               constraintsVSMPC.cpp:85-103): Householder QR of (Lambda W^-1/2)^T, W the joint weights (costsVSMPC.cpp:
 *               375-381,564-591) -> 6 unknowns y per block with input matrix R^T, unit weights, gradient Q^T b; the
 *               2-dimensional null component is closed form (kernel v24, tests/algo_model.py: joint_reduction)
 *   condense    sensitivity recursion S_{k+1} = (I + dt A) S_k + dt E_k over the condensed columns
 *               [y_0..y_{H-1} | v_1..v_{nvb-1} | v_0 | affine], C += Y^T Y with Y = sqrt(Q) S on the 18 weighted rows
 *               (constraintsVSMPC.cpp:76-131, costsVSMPC.cpp:166-200)
 *   augment     + unit joint weights, throttle first-difference penalty and anchor, gradient row (costsVSMPC.cpp:375-409,468-487,558-592)
 *   factor      dense Cholesky of the NZ x NZ Hessian, the gradient row carried along
 *   box QP      block principal pivoting on the Schur complement of the throttles (constraintsVSMPC.cpp:338-365),
 *               least-index fallback, as tests/algo_model.py
 *   back-subst  reduced joints; forward simulation for the state trajectory (variableSamplingMPC.cpp:93-108); U = W^-1/2 (Q y + N n)
 *
 * Linked into oracle/_build/liboracle.so together with vsmpc_oracle.c.
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
#define NJC 6   /* reduced joint unknowns per block */
#define NT 4

typedef struct vso_config {
    int n_iter, n_iter_small, control_horizon, use_jet_dynamic;
    double period_mpc, period_small, period_large;
    double w_com_pos[3], w_com_pos_err[3], w_lin_mom[3], w_rpy[3], w_rpy_err[3], w_ang_mom[3];
    double w_delta_joint[8];
    double w_throttle, w_initial_throttle, w_reg_joint_pos, throttle_min, throttle_max;
} vso_config;

void vso_dt_schedule(const vso_config* c, double* dt);
void vso_linearize(const vso_config* c, const double* in, double* A, double* Bj, double* Bt, double* cv);

enum { IN_X0 = 0, IN_UPREV = 141, IN_QERR = 153, IN_HOLD = 161, IN_XREF = 162 };
static const double JC12 = -7.83052261e-03, JNU0 = 47.333, JNU1 = 31.483;
static double v_of_throttle(double u) { const double ub = (u - JNU0) / JNU1; return ub + JC12 * ub * ub; }

static const int WROW[18] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 20, 21, 22, 23, 24, 25};

/* x (reference variable order), returns status 1 solved / 2 iteration cap / 3 not positive definite; *iters = active-set
 * iterations.  Work arrays are allocated per call (small next to the arithmetic). */
int vss_solve(const vso_config* c, const double* in, double* x, int* iters) {
    const int N = c->n_iter, nS = c->n_iter_small, H = c->control_horizon, nvb = H - nS + 1;
    const int NU = NJC * H, NV = NT * nvb, NZ = NU + NV, NC = NZ + 1;   /* + the affine column */
    const int nref = N - nS + 1;
    double A[NX * NX], Bj8[NX * NJ], Bj[NX * NJC], Bt[NX * NT], cv[NX], dt[64], q[NX], sq[18];
    vso_linearize(c, in, A, Bj8, Bt, cv);
    vso_dt_schedule(c, dt);
    /* joint reduction: columns of Ac = (Lambda W^-1/2)^T (8 x 6), b = w_reg W^-1/2 q_err rides along */
    double Ac[8][NJC], bt[8], Vh[NJC][8], beta[NJC], isw[8];
    for (int j = 0; j < 8; ++j) {
        isw[j] = 1.0 / sqrt(c->w_delta_joint[j] + c->w_reg_joint_pos);
        for (int a = 0; a < NJC; ++a) Ac[j][a] = Bj8[((a < 3 ? 3 : 6) + a) * NJ + j] * isw[j];
        bt[j] = c->w_reg_joint_pos * in[IN_QERR + j] * isw[j];
    }
    for (int k = 0; k < NJC; ++k) {
        double s2 = 0.0;
        for (int i = k; i < 8; ++i) { s2 += Ac[i][k] * Ac[i][k]; Vh[k][i] = 0.0; }
        for (int i = 0; i < k; ++i) Vh[k][i] = 0.0;
        beta[k] = 0.0;
        if (!(s2 > 1e-300)) continue;
        const double nrm = sqrt(s2), x0 = Ac[k][k], alpha = x0 >= 0.0 ? -nrm : nrm;
        for (int i = k; i < 8; ++i) Vh[k][i] = Ac[i][k];
        Vh[k][k] = x0 - alpha;
        beta[k] = 1.0 / (nrm * (nrm + fabs(x0)));
        for (int col = k; col < NJC; ++col) {
            double w = 0.0;
            for (int i = k; i < 8; ++i) w += Vh[k][i] * Ac[i][col];
            w *= beta[k];
            for (int i = k; i < 8; ++i) Ac[i][col] -= w * Vh[k][i];
        }
        double w = 0.0;
        for (int i = k; i < 8; ++i) w += Vh[k][i] * bt[i];
        w *= beta[k];
        for (int i = k; i < 8; ++i) bt[i] -= w * Vh[k][i];
    }
    memset(Bj, 0, sizeof(Bj));   /* input matrix of the reduced unknowns: R^T on the momentum rows */
    for (int a = 0; a < NJC; ++a)
        for (int k = 0; k <= a; ++k) Bj[((a < 3 ? 3 : 6) + a) * NJC + k] = Ac[k][a];
    memset(q, 0, sizeof(q));
    for (int i = 0; i < 3; ++i) {
        q[i] = c->w_com_pos[i]; q[3 + i] = c->w_lin_mom[i]; q[6 + i] = c->w_rpy[i]; q[9 + i] = c->w_ang_mom[i];
        q[20 + i] = c->w_com_pos_err[i]; q[23 + i] = c->w_rpy_err[i];
    }
    for (int i = 0; i < 18; ++i) sq[i] = sqrt(q[WROW[i]]);
    /* column kinds: joint (block, comp), throttle (reference block, comp) in the internal order v_1..v_{nvb-1}, v_0 */
    int* vblk = (int*)malloc(sizeof(int) * NV);
    for (int k = 0; k < NV; ++k) { const int b = k / NT; vblk[k] = b < nvb - 1 ? b + 1 : 0; }
    double* S = (double*)calloc((size_t)NX * NC, sizeof(double));
    double* Sn = (double*)malloc(sizeof(double) * NX * NC);
    double* Y = (double*)malloc(sizeof(double) * 18 * NC);
    double* M = (double*)calloc((size_t)NC * NC, sizeof(double));
    for (int r = 0; r < NX; ++r) S[r * NC + NZ] = in[IN_X0 + r];
    for (int k = 0; k < N; ++k) {
        const int jb = k < H ? k : H - 1;
        const int tb = k < nS ? 0 : (k < H ? k - (nS - 1) : H - nS);
        /* Sn = S + dt (A S + E) */
        for (int r = 0; r < NX; ++r) {
            double* o = Sn + (size_t)r * NC;
            for (int col = 0; col < NC; ++col) o[col] = 0.0;
            for (int t = 0; t < NX; ++t) {
                const double a = A[r * NX + t];
                if (a == 0.0) continue;
                const double* s = S + (size_t)t * NC;
                for (int col = 0; col < NC; ++col) o[col] += a * s[col];
            }
            for (int j = 0; j < NJC; ++j) o[NJC * jb + j] += Bj[r * NJC + j];
            for (int kk = 0; kk < NV; ++kk) if (vblk[kk] == tb) o[NU + kk] += Bt[r * NT + (kk % NT)];
            o[NZ] += cv[r];
            const double* s0 = S + (size_t)r * NC;
            for (int col = 0; col < NC; ++col) o[col] = s0[col] + dt[k] * o[col];
        }
        { double* t = S; S = Sn; Sn = t; }
        const int rc = k < nS ? 0 : k - nS;
        const double* xr = in + IN_XREF + 12 * rc;
        for (int i = 0; i < 18; ++i) {
            const double* s = S + (size_t)WROW[i] * NC;
            double* y = Y + (size_t)i * NC;
            for (int col = 0; col < NC; ++col) y[col] = sq[i] * s[col];
            if (WROW[i] < 12) y[NZ] -= sq[i] * xr[WROW[i]];
        }
        for (int i = 0; i < 18; ++i) {   /* lower triangle of M += Y^T Y */
            const double* y = Y + (size_t)i * NC;
            for (int r = 0; r < NC; ++r) {
                const double yr = y[r];
                if (yr == 0.0) continue;
                double* m = M + (size_t)r * NC;
                for (int col = 0; col <= r; ++col) m[col] += yr * y[col];
            }
        }
    }
    (void)nref;
    /* augment */
    double vprev[NT];
    for (int i = 0; i < NT; ++i) vprev[i] = v_of_throttle(in[IN_UPREV + i]);
    for (int col = 0; col < NU; ++col) {
        M[(size_t)col * NC + col] += 1.0;                    /* |y|^2 / 2 = U^T W U / 2 in the reduced unknowns */
        M[(size_t)NZ * NC + col] += bt[col % NJC];           /* Q^T b */
    }
    for (int a = 0; a < NV; ++a)
        for (int b = 0; b <= a; ++b) {
            if (a % NT != b % NT) continue;
            const int ba = vblk[a], bb = vblk[b];
            double w = 0.0;
            if (ba == bb) w = c->w_throttle * ((ba > 0) + (ba < nvb - 1)) + (ba == 0 ? c->w_initial_throttle : 0.0);
            else if (ba - bb == 1 || bb - ba == 1) w = -c->w_throttle;
            M[(size_t)(NU + a) * NC + NU + b] += w;
        }
    for (int k = 0; k < NV; ++k) if (vblk[k] == 0) M[(size_t)NZ * NC + NU + k] += -c->w_initial_throttle * vprev[k % NT];
    /* Cholesky of the leading NZ x NZ block, row NZ carried along (becomes (L^-1 g)^T) */
    int status = 1;
    for (int j = 0; j < NZ && status == 1; ++j) {
        double d = M[(size_t)j * NC + j];
        for (int t = 0; t < j; ++t) d -= M[(size_t)j * NC + t] * M[(size_t)j * NC + t];
        if (!(d > 0.0)) { status = 3; break; }
        const double l = sqrt(d), il = 1.0 / l;
        M[(size_t)j * NC + j] = l;
        for (int r = j + 1; r <= NZ; ++r) {
            double s = M[(size_t)r * NC + j];
            const double* mr = M + (size_t)r * NC;
            const double* mj = M + (size_t)j * NC;
            for (int t = 0; t < j; ++t) s -= mr[t] * mj[t];
            M[(size_t)r * NC + j] = s * il;
        }
    }
    int it = 0;
    double* z = (double*)calloc(NZ, sizeof(double));
    if (status == 1) {
        /* box QP on the throttles: S = L22 L22^T, s = L22 ghat_v */
        double Sv[64 * 64], sv[64], lo[64], hi[64], v[64];
        int state[64], fixed[64];
        const double vmin = v_of_throttle(c->throttle_min), vmax = v_of_throttle(c->throttle_max);
        const int hold = in[IN_HOLD] != 0.0;
        for (int a = 0; a < NV; ++a) {
            for (int b = 0; b < NV; ++b) {
                double s = 0.0;
                const int kmax = a < b ? a : b;
                for (int t = 0; t <= kmax; ++t) s += M[(size_t)(NU + a) * NC + NU + t] * M[(size_t)(NU + b) * NC + NU + t];
                Sv[a * 64 + b] = s;
            }
            double s = 0.0;
            for (int t = 0; t <= a; ++t) s += M[(size_t)(NU + a) * NC + NU + t] * M[(size_t)NZ * NC + NU + t];
            sv[a] = s;
            fixed[a] = hold && a >= NV - NT;
            lo[a] = fixed[a] ? vprev[a % NT] : vmin;
            hi[a] = fixed[a] ? vprev[a % NT] : vmax;
            state[a] = fixed[a] ? -1 : 0;
        }
        double gmax = 0.0;
        for (int a = 0; a < NV; ++a) if (fabs(sv[a]) > gmax) gmax = fabs(sv[a]);
        const double gtol = 1e-10 * (1.0 + gmax);
        int best = NV + 1, patience = 10;
        status = 2;
        for (it = 1; it <= 64; ++it) {
            /* masked system: identity rows for bound throttles */
            double K[64 * 64], rhs[64];
            for (int a = 0; a < NV; ++a) {
                const int Fa = state[a] == 0;
                double r = Fa ? -sv[a] : (state[a] < 0 ? lo[a] : hi[a]);
                for (int b = 0; b < NV; ++b) {
                    const int Fb = state[b] == 0;
                    K[a * 64 + b] = (Fa && Fb) ? Sv[a * 64 + b] : (a == b && !Fa ? 1.0 : 0.0);
                    if (Fa && !Fb) r -= Sv[a * 64 + b] * (state[b] < 0 ? lo[b] : hi[b]);
                }
                rhs[a] = r;
            }
            int bad = 0;
            for (int j = 0; j < NV; ++j) {   /* Cholesky-free symmetric elimination (SPD on the free block) */
                const double piv = K[j * 64 + j];
                if (!(piv > 0.0)) { bad = 1; break; }
                for (int r = j + 1; r < NV; ++r) {
                    const double f = K[r * 64 + j] / piv;
                    if (f == 0.0) continue;
                    for (int cc = j; cc < NV; ++cc) K[r * 64 + cc] -= f * K[j * 64 + cc];
                    rhs[r] -= f * rhs[j];
                }
            }
            if (bad) { status = 3; break; }
            for (int j = NV - 1; j >= 0; --j) {
                double s = rhs[j];
                for (int cc = j + 1; cc < NV; ++cc) s -= K[j * 64 + cc] * v[cc];
                v[j] = s / K[j * 64 + j];
            }
            int ninf = 0, inf[64], kind[64];
            for (int a = 0; a < NV; ++a) {
                double grad = sv[a];
                for (int b = 0; b < NV; ++b) grad += Sv[a * 64 + b] * v[b];
                const double tolv = 1e-12 * (1.0 + fabs(v[a]));
                inf[a] = 0; kind[a] = 0;
                if (state[a] == 0 && v[a] < lo[a] - tolv) { inf[a] = 1; kind[a] = -1; }
                else if (state[a] == 0 && v[a] > hi[a] + tolv) { inf[a] = 1; kind[a] = 1; }
                else if (state[a] == -1 && !fixed[a] && grad < -gtol) { inf[a] = 1; kind[a] = 0; }
                else if (state[a] == 1 && !fixed[a] && grad > gtol) { inf[a] = 1; kind[a] = 0; }
                ninf += inf[a];
            }
            if (ninf == 0) { status = 1; break; }
            int single = -1;
            if (ninf < best) { best = ninf; patience = 10; }
            else if (patience > 0) --patience;
            else for (int a = 0; a < NV; ++a) if (inf[a]) single = a;   /* largest index */
            for (int a = 0; a < NV; ++a) if (inf[a] && (single < 0 || a == single)) state[a] = kind[a];
        }
        for (int a = 0; a < NV; ++a) z[NU + a] = state[a] < 0 ? lo[a] : (state[a] > 0 ? hi[a] : v[a]);
        /* joints: L11^T U = y_U - L21^T v, y = -ghat */
        for (int j = NU - 1; j >= 0; --j) {
            double s = -M[(size_t)NZ * NC + j];
            for (int r = j + 1; r < NZ; ++r) s -= M[(size_t)r * NC + j] * z[r];
            z[j] = s / M[(size_t)j * NC + j];
        }
    }
    /* forward simulation, output in the reference's variable order */
    double X[NX];
    for (int r = 0; r < NX; ++r) { X[r] = in[IN_X0 + r]; x[r] = X[r]; }
    for (int k = 0; k < N; ++k) {
        const int jb = k < H ? k : H - 1;
        const int tb = k < nS ? 0 : (k < H ? k - (nS - 1) : H - nS);
        const int vq = tb == 0 ? NV - NT : NT * (tb - 1);
        double Xn[NX];
        for (int r = 0; r < NX; ++r) {
            double f = cv[r];
            for (int t = 0; t < NX; ++t) f += A[r * NX + t] * X[t];
            for (int j = 0; j < NJC; ++j) f += Bj[r * NJC + j] * z[NJC * jb + j];
            for (int j = 0; j < NT; ++j) f += Bt[r * NT + j] * z[NU + vq + j];
            Xn[r] = X[r] + dt[k] * f;
        }
        memcpy(X, Xn, sizeof(X));
        memcpy(x + NX * (k + 1), X, sizeof(X));
    }
    for (int i = 0; i < H; ++i) {   /* U_i = W^-1/2 H_1 .. H_6 [y_i; n], n = -N^T b */
        double u[8];
        for (int k = 0; k < NJC; ++k) u[k] = z[NJC * i + k];
        u[6] = -bt[6]; u[7] = -bt[7];
        for (int k = NJC - 1; k >= 0; --k) {
            double w = 0.0;
            for (int t = k; t < 8; ++t) w += Vh[k][t] * u[t];
            w *= beta[k];
            for (int t = k; t < 8; ++t) u[t] -= w * Vh[k][t];
        }
        for (int k = 0; k < 8; ++k) x[NX * (N + 1) + NJ * i + k] = u[k] * isw[k];
    }
    for (int b = 0; b < nvb; ++b)
        for (int j = 0; j < NT; ++j) x[NX * (N + 1) + NJ * H + NT * b + j] = z[NU + (b == 0 ? NV - NT : NT * (b - 1)) + j];
    if (iters) *iters = it;
    free(z); free(M); free(Y); free(S); free(Sn); free(vblk);
    return status;
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

/* like vso_time_batch: `threads` workers, stops after budget_s; stats[0] = mean iterations, stats[2] = solved fraction */
double vss_time_batch(const vso_config* c, const double* in, int batch, int threads, double budget_s, int* done,
                      double* x_out, double* stats) {
    const int N = c->n_iter, H = c->control_horizon, nvb = H - c->n_iter_small + 1;
    const int nv = NX * (N + 1) + NJ * H + NT * nvb, nin = IN_XREF + 12 * (N - c->n_iter_small + 1);
    if (threads < 1) threads = 1;
    int ndone = 0, sum_ok = 0;
    long sum_it = 0;
    const double t0 = now_s();
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1) reduction(+ : ndone, sum_it, sum_ok)
    for (int b = 0; b < batch; ++b) {
        if (now_s() - t0 > budget_s) continue;
        double* xb = (double*)malloc(sizeof(double) * nv);
        int it = 0;
        const int st = vss_solve(c, in + (size_t)b * nin, xb, &it);
        if (x_out) memcpy(x_out + (size_t)b * nv, xb, sizeof(double) * nv);
        free(xb);
        ndone += 1; sum_it += it; sum_ok += (st == 1);
    }
    const double el = now_s() - t0;
    *done = ndone;
    if (stats && ndone > 0) { stats[0] = (double)sum_it / ndone; stats[1] = 0.0; stats[2] = (double)sum_ok / ndone; }
    return el;
}
