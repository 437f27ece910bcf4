// Batched kinematics provider: what the reference's Robot::setState caches for the MPC path
// (utils/src/Robot.cpp:198-335; getJacobian :505-514), for a simplified tree committed as plain arrays (`vsmpc_tree`,
// include/vsmpc.h): floating base + 8 revolute joints + 4 jet frames.  One wavefront per instance:
//   forward kinematics along the (short) chains, body velocities                       lanes = chains from the base
//   total mass, CoM, centroidal momentum, base block of the free-floating mass matrix  (MIXED representation, iDynTree)
//   jet axes / arms, A_mom (world and body coordinates)                                lanes = jets
//   frame, CoM and relative Jacobians, written straight into the VSMPC_KIN_* record    lanes = (jet, joint) pairs
// and, when asked to, the VSMPC_IN_* fields update() would pull out of the Robot (X0 position / momentum / RPY / thrusts,
// mass, wR_b, omega_B, gravity, A_mom,body), so that provider -> vsmpc_kinematics -> solve runs without the host.
// HBM-bound and tiny (38 doubles in, ~0.8k doubles out per instance).  Oracle: oracle/robot_tree_ref.py (parity
// unpinned: iDynTree and the URDF are not in this image; the oracle is pinned by finite differences and energy checks).
#include "vsmpc_device.hpp"
#include "vsmpc_launch.hpp"

namespace vsmpc {

namespace {

constexpr int NB = VSMPC_TREE_NB, NJT = VSMPC_TREE_NJ, NJETS = VSMPC_N_THRUSTS, NJR = VSMPC_KIN_NJ;

__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void matvec(const double* R, const double* v, double* o) {   // R v, row-major
    for (int i = 0; i < 3; ++i) o[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}
__device__ __forceinline__ void matTvec(const double* R, const double* v, double* o) {  // R^T v
    for (int i = 0; i < 3; ++i) o[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2];
}

}  // namespace

__global__ __launch_bounds__(64) void provider_kernel(vsmpc_tree tree, const double* __restrict__ state, int batch,
                                                      double* __restrict__ kin, double* __restrict__ robot,
                                                      double* __restrict__ records, int n_in) {
    __shared__ double sR[NB][9], sP[NB][3], sAx[NJT][3], sOrg[NJT][3], sC[NB][3], sI[NB][9], sW[NB][3], sV[NB][3];
    __shared__ double sTot[32];   // mass, com[3], h_lin[3], h_ang[3], Mb is written straight out
    __shared__ double sJet[NJETS][9];   // position, axis, arm
    __shared__ int sAnc[NB];            // joints between the base and the body, as a bit mask
    const int b0 = blockIdx.x, lane = threadIdx.x;
    if (b0 >= batch) return;
    const double* s = state + size_t(b0) * VSMPC_RS_SIZE;
    double* K = kin + size_t(b0) * VSMPC_KIN_SIZE;
    for (int i = lane; i < VSMPC_KIN_SIZE; i += 64) K[i] = 0.0;   // Jacobian columns of unmodelled joints stay zero
    // ---- forward kinematics: bodies in index order (parents precede children), one lane; the chains are 4 joints deep
    if (lane == 0) {
        for (int i = 0; i < 9; ++i) sR[0][i] = s[VSMPC_RS_R + i];
        for (int i = 0; i < 3; ++i) { sP[0][i] = s[VSMPC_RS_P + i]; sW[0][i] = s[VSMPC_RS_W + i]; }
        sAnc[0] = 0;
        for (int j = 0; j < NJT; ++j) {
            const int b = j + 1, par = tree.parent[b];
            double ax[3] = {tree.joint_axis[3 * j], tree.joint_axis[3 * j + 1], tree.joint_axis[3 * j + 2]};
            const double nrm = rsqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
            for (int i = 0; i < 3; ++i) ax[i] *= nrm;
            double o[3];
            matvec(sR[par], &tree.joint_origin[3 * j], o);
            for (int i = 0; i < 3; ++i) { sOrg[j][i] = sP[par][i] + o[i]; sP[b][i] = sOrg[j][i]; }
            matvec(sR[par], ax, sAx[j]);
            // Rodrigues: Rj = I + sin K + (1 - cos) K^2
            double sn, cs;
            sincos(s[VSMPC_RS_Q + j], &sn, &cs);
            const double Kx[9] = {0.0, -ax[2], ax[1], ax[2], 0.0, -ax[0], -ax[1], ax[0], 0.0};
            double Rj[9];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    double k2 = 0.0;
                    for (int t = 0; t < 3; ++t) k2 += Kx[3 * r + t] * Kx[3 * t + c];
                    Rj[3 * r + c] = (r == c ? 1.0 : 0.0) + sn * Kx[3 * r + c] + (1.0 - cs) * k2;
                }
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    double v = 0.0;
                    for (int t = 0; t < 3; ++t) v += sR[par][3 * r + t] * Rj[3 * t + c];
                    sR[b][3 * r + c] = v;
                }
            sAnc[b] = sAnc[par] | (1 << j);
            const double qd = s[VSMPC_RS_QD + j];
            for (int i = 0; i < 3; ++i) sW[b][i] = sW[par][i] + qd * sAx[j][i];
        }
    }
    __syncthreads();
    // ---- per body (lane = body): world CoM, world inertia, CoM velocity
    if (lane < NB) {
        const int b = lane;
        double c[3];
        matvec(sR[b], &tree.com[3 * b], c);
        for (int i = 0; i < 3; ++i) sC[b][i] = sP[b][i] + c[i];
        const double* in = &tree.inertia[6 * b];
        const double Ib[9] = {in[0], in[1], in[2], in[1], in[3], in[4], in[2], in[4], in[5]};
        double RI[9];
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc) {
                double v = 0.0;
                for (int t = 0; t < 3; ++t) v += sR[b][3 * r + t] * Ib[3 * t + cc];
                RI[3 * r + cc] = v;
            }
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc) {
                double v = 0.0;
                for (int t = 0; t < 3; ++t) v += RI[3 * r + t] * sR[b][3 * cc + t];
                sI[b][3 * r + cc] = v;
            }
        // v_c = v_base + w_base x (c - p_base) + sum_{j upstream} qd_j axis_j x (c - o_j)
        double d[3], x[3], v[3];
        for (int i = 0; i < 3; ++i) d[i] = sC[b][i] - sP[0][i];
        cross3(sW[0], d, x);
        for (int i = 0; i < 3; ++i) v[i] = s[VSMPC_RS_V + i] + x[i];
        for (int j = 0; j < NJT; ++j)
            if (sAnc[b] & (1 << j)) {
                for (int i = 0; i < 3; ++i) d[i] = sC[b][i] - sOrg[j][i];
                cross3(sAx[j], d, x);
                const double qd = s[VSMPC_RS_QD + j];
                for (int i = 0; i < 3; ++i) v[i] += qd * x[i];
            }
        for (int i = 0; i < 3; ++i) sV[b][i] = v[i];
    }
    __syncthreads();
    // ---- totals (one lane; nine bodies)
    if (lane == 0) {
        double m = 0.0, com[3] = {0, 0, 0}, hl[3] = {0, 0, 0}, ha[3] = {0, 0, 0}, IO[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int b = 0; b < NB; ++b) {
            m += tree.mass[b];
            for (int i = 0; i < 3; ++i) { com[i] += tree.mass[b] * sC[b][i]; hl[i] += tree.mass[b] * sV[b][i]; }
        }
        for (int i = 0; i < 3; ++i) com[i] /= m;
        for (int b = 0; b < NB; ++b) {
            double Iw[3], d[3], x[3];
            matvec(sI[b], sW[b], Iw);
            for (int i = 0; i < 3; ++i) d[i] = sC[b][i] - com[i];
            cross3(d, sV[b], x);
            for (int i = 0; i < 3; ++i) ha[i] += Iw[i] + tree.mass[b] * x[i];
            for (int i = 0; i < 3; ++i) d[i] = sC[b][i] - sP[0][i];
            const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) IO[3 * r + c] += sI[b][3 * r + c] + tree.mass[b] * ((r == c ? dd : 0.0) - d[r] * d[c]);
        }
        sTot[0] = m;
        for (int i = 0; i < 3; ++i) { sTot[1 + i] = com[i]; sTot[4 + i] = hl[i]; sTot[7 + i] = ha[i]; }
        // base block of the free-floating mass matrix (MIXED): [[m I, -m S(c)], [m S(c), I_O]], c = com - p_base
        const double c[3] = {com[0] - sP[0][0], com[1] - sP[0][1], com[2] - sP[0][2]};
        const double S[9] = {0.0, -c[2], c[1], c[2], 0.0, -c[0], -c[1], c[0], 0.0};
        double* Mb = K + VSMPC_KIN_MB;
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc) {
                Mb[6 * r + cc] = r == cc ? m : 0.0;
                Mb[6 * r + 3 + cc] = -m * S[3 * r + cc];
                Mb[6 * (3 + r) + cc] = m * S[3 * r + cc];
                Mb[6 * (3 + r) + 3 + cc] = IO[3 * r + cc];
            }
        for (int i = 0; i < 3; ++i) K[VSMPC_KIN_R + i] = c[i];
        for (int i = 0; i < 9; ++i) K[VSMPC_KIN_WRB + i] = sR[0][i];
        for (int i = 0; i < NJETS; ++i) K[VSMPC_KIN_THRUST + i] = s[VSMPC_RS_T + i];
    }
    __syncthreads();
    // ---- jets (lane = jet): position, axis (Robot.cpp:256), arm (:258, zero CoM offset)
    if (lane < NJETS) {
        const int i = lane, b = tree.jet_body[i];
        double o[3], a[3] = {tree.jet_axis[3 * i], tree.jet_axis[3 * i + 1], tree.jet_axis[3 * i + 2]};
        const double nrm = rsqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        for (int t = 0; t < 3; ++t) a[t] *= nrm;
        matvec(sR[b], &tree.jet_origin[3 * i], o);
        matvec(sR[b], a, &sJet[i][3]);
        for (int t = 0; t < 3; ++t) {
            sJet[i][t] = sP[b][t] + o[t];
            sJet[i][6 + t] = sJet[i][t] - sTot[1 + t];
            K[VSMPC_KIN_AXES + 3 * i + t] = sJet[i][3 + t];
            K[VSMPC_KIN_ARMS + 3 * i + t] = sJet[i][6 + t];
        }
    }
    __syncthreads();
    // ---- Jacobian columns: lane = (jet i, joint j); the CoM Jacobian in lanes 32..39 (lane - 32 = joint)
    if (lane < NJETS * NJT) {
        const int i = lane >> 3, j = lane & 7, col = tree.robot_joint[j];
        if (sAnc[tree.jet_body[i]] & (1 << j)) {
            double d[3], lin[3], rel[3];
            for (int t = 0; t < 3; ++t) d[t] = sJet[i][t] - sOrg[j][t];
            cross3(sAx[j], d, lin);                       // getFrameFreeFloatingJacobian, linear rows, joint column
            matTvec(sR[0], sAx[j], rel);                  // getRelativeJacobian(base, jet), angular rows, base axes
            for (int t = 0; t < 3; ++t) {
                K[VSMPC_KIN_JFRAME + (i * 3 + t) * NJR + col] = lin[t];
                K[VSMPC_KIN_JREL + (i * 3 + t) * NJR + col] = rel[t];
            }
        }
    } else if (lane < NJETS * NJT + NJT) {
        const int j = lane - NJETS * NJT, col = tree.robot_joint[j];
        double acc[3] = {0, 0, 0};
        for (int b = 0; b < NB; ++b)
            if (sAnc[b] & (1 << j)) {
                double d[3], x[3];
                for (int t = 0; t < 3; ++t) d[t] = sC[b][t] - sOrg[j][t];
                cross3(sAx[j], d, x);
                for (int t = 0; t < 3; ++t) acc[t] += tree.mass[b] * x[t];
            }
        for (int t = 0; t < 3; ++t) K[VSMPC_KIN_JCOM + t * NJR + col] = acc[t] / sTot[0];
    }
    // ---- Robot-level outputs (lane 0) and the record fields update() pulls out of the Robot
    if (lane == 0) {
        double hb[6], rpy[3];
        matTvec(sR[0], &sTot[4], hb);
        matTvec(sR[0], &sTot[7], hb + 3);                 // Robot.cpp:324-326
        const double* R = sR[0];
        rpy[0] = atan2(R[7], R[8]);
        rpy[1] = atan2(-R[6], hypot(R[7], R[8]));
        rpy[2] = atan2(R[3], R[0]);
        double Am[24], Ab[24];
        for (int i = 0; i < NJETS; ++i) {
            double ra[3], t1[3], t2[3];
            cross3(&sJet[i][6], &sJet[i][3], ra);         // S(r) a  (:263-264)
            matTvec(R, &sJet[i][3], t1);
            matTvec(R, ra, t2);                           // :327-328
            for (int t = 0; t < 3; ++t) {
                Am[4 * t + i] = sJet[i][3 + t];
                Am[4 * (3 + t) + i] = ra[t];
                Ab[4 * t + i] = t1[t];
                Ab[4 * (3 + t) + i] = t2[t];
            }
        }
        if (robot != nullptr) {
            double* o = robot + size_t(b0) * VSMPC_RO_SIZE;
            for (int i = 0; i < 3; ++i) { o[VSMPC_RO_COM + i] = sTot[1 + i]; o[VSMPC_RO_RPY + i] = rpy[i]; }
            for (int i = 0; i < 3; ++i) { o[VSMPC_RO_MOM + i] = sTot[4 + i]; o[VSMPC_RO_MOM + 3 + i] = sTot[7 + i]; }
            for (int i = 0; i < 6; ++i) o[VSMPC_RO_MOMB + i] = hb[i];
            o[VSMPC_RO_MASS] = sTot[0];
            for (int i = 0; i < 24; ++i) { o[VSMPC_RO_AMOM + i] = Am[i]; o[VSMPC_RO_AMOMB + i] = Ab[i]; }
        }
        if (records != nullptr) {
            double* rec = records + size_t(b0) * n_in;
            for (int i = 0; i < 3; ++i) {
                rec[VSMPC_IN_X0 + i] = sTot[1 + i];       // constraintsVSMPC.cpp:209-216
                rec[VSMPC_IN_X0 + 3 + i] = hb[i];
                rec[VSMPC_IN_X0 + 6 + i] = rpy[i];        // (not unwrapped: the tick state machine owns the turn counters)
                rec[VSMPC_IN_X0 + 9 + i] = hb[3 + i];
                rec[VSMPC_IN_RPY + i] = rpy[i];
                rec[VSMPC_IN_GRAV + i] = tree.gravity[i];
            }
            double wB[3];
            matTvec(R, sW[0], wB);                        // systemDynamicsVSMPC.cpp:108,325
            for (int i = 0; i < 3; ++i) rec[VSMPC_IN_OMEGA + i] = wB[i];
            for (int i = 0; i < 9; ++i) rec[VSMPC_IN_WRB + i] = R[i];
            rec[VSMPC_IN_MASS] = double(float(sTot[0]));  // Robot.h:338 keeps a float
            for (int i = 0; i < 24; ++i) rec[VSMPC_IN_AMOM + i] = Ab[i];
            for (int i = 0; i < NJETS; ++i) {
                rec[VSMPC_IN_X0 + 12 + i] = s[VSMPC_RS_T + i];
                rec[VSMPC_IN_T0 + i] = s[VSMPC_RS_T + i];
            }
        }
    }
}

hipError_t launch_provider(const vsmpc_tree& tree, const double* d_state, int batch, double* d_kin, double* d_robot,
                           double* d_records, int n_in, hipStream_t stream) {
    hipLaunchKernelGGL(provider_kernel, dim3(batch), dim3(64), 0, stream, tree, d_state, batch, d_kin, d_robot, d_records, n_in);
    return hipGetLastError();
}

}  // namespace vsmpc
