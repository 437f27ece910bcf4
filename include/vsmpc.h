/*
 * vsmpc.h — C-ABI of the MI355X-native batched multi-rate ("variable sampling") MPC solve path.
 *
 * Drop-in boundary for ONE hot path of ami-iit/paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub:
 * the per-tick  IMPCProblem::update() + VariableSamplingMPC::solveMPC()  pair
 *   (momentum-based-linear-mpc-lib/src/IMPCProblem/IMPCProblem.cpp:150-298,
 *    momentum-based-linear-mpc-lib/src/variableSamplingMPC/variableSamplingMPC.cpp:88-112),
 * i.e. linearise -> variable-sampling QP assembly -> QP solve -> first-move extraction, with a batch
 * axis over independent MPC instances added.  Plain C: pointers and sizes only, no C++/torch types.
 *
 * Each entry point cites the reference interface it replaces (paths relative to
 * /root/reference/src/flight-controller/).  INTEGRATION.md shows the reference-side binding.
 *
 * Threading: one handle per host thread / HIP stream; a handle is not thread-safe, distinct handles
 * are independent (the reference object is single-threaded and stateful too).
 * Errors: int return, 0 = VSMPC_OK, negative = API/HIP error (never throws, never aborts), plus a
 * per-instance status array mirroring the OsqpEigen::Status values the reference distinguishes
 * (IMPCProblem.cpp:285-294): the caller applies "consume only if Solved" (variableSamplingMPC.cpp:91).
 */
#ifndef VSMPC_H
#define VSMPC_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- sizes fixed by the reference (variableSamplingMPC/VSconstant.h:6-16) ---- */
#define VSMPC_N_STATES 26
#define VSMPC_N_JOINTS 8
#define VSMPC_N_THRUSTS 4

/* ---- return codes ---- */
#define VSMPC_OK 0
#define VSMPC_ERR_INVALID_ARG (-1)
#define VSMPC_ERR_UNSUPPORTED_CONFIG (-2)
#define VSMPC_ERR_BATCH_TOO_LARGE (-3)
#define VSMPC_ERR_HIP (-4)
#define VSMPC_ERR_ALLOC (-5)

/* ---- per-instance status (IMPCProblem.cpp:285-294) ---- */
#define VSMPC_STATUS_SOLVED 1        /* exact optimum found: OsqpEigen::Status::Solved          */
#define VSMPC_STATUS_MAX_ITER 2      /* active-set iteration cap hit: ...::MaxIterReached        */
#define VSMPC_STATUS_NUMERICAL 3     /* non-positive pivot / non-finite data: ...::NonCvx/error  */

/*
 * Configuration = the keys of group VS_MPC_CONFIG (src/config/vs_mcp_config.xml:7-43) that the path
 * reads (variableSamplingMPC.cpp:24-38, constraintsVSMPC.cpp:25-51,295-322, costsVSMPC.cpp:29-69,
 * 326-361,445-459,516-535).  XML/TOML parsing stays on the caller's side.
 */
typedef struct vsmpc_config {
    int n_iter;            /* nIter                */
    int n_iter_small;      /* nIterSmall           */
    int control_horizon;   /* controlHorizon       */
    int use_jet_dynamic;   /* useJetDynamic        */
    double period_mpc;     /* periodMPC            */
    double period_small;   /* periodMPCSmallSteps  */
    double period_large;   /* periodMPCLargeSteps  */
    double w_com_pos[3];       /* weightCoMPos      */
    double w_com_pos_err[3];   /* weightCoMPosError */
    double w_lin_mom[3];       /* weightLinMom      */
    double w_rpy[3];           /* weightRPY         */
    double w_rpy_err[3];       /* weightRPYError    */
    double w_ang_mom[3];       /* weightAngMom      */
    double w_delta_joint[8];   /* weightDeltaJoint  */
    double w_throttle;         /* weightThrottle    */
    double w_initial_throttle; /* weightInitialThrottle */
    double w_reg_joint_pos;    /* weightRegularizationJointPos */
    double throttle_min;       /* throttleMin (percent) */
    double throttle_max;       /* throttleMax (percent) */
} vsmpc_config;

/*
 * Per-instance input record: `vsmpc_input_doubles()` doubles, array-of-records, C-contiguous
 * ([batch][n_in]).  It is the data the reference's update() pulls out of QPInput/Robot each tick
 * (utils/include/QPInput.h:12-124; consumers cited per field).  Offsets in doubles:
 */
#define VSMPC_IN_X0 0         /* 26 measured state X0: CoM, h_lin(body), unwrapped RPY, h_ang(body), T, Tdot,
                                    CoM - posCoMReference, rpy - RPYReference (constraintsVSMPC.cpp:206-230) */
#define VSMPC_IN_MASS 26      /*  1 Robot::getTotalMass (a float in the reference, utils/include/Robot.h:338) */
#define VSMPC_IN_WRB 27       /*  9 base rotation wR_b, row-major (systemDynamicsVSMPC.cpp:107,324)            */
#define VSMPC_IN_OMEGA 36     /*  3 omega_B = wR_b^T * base angular velocity (systemDynamicsVSMPC.cpp:108,325) */
#define VSMPC_IN_ALPHA 39     /*  1 alpha_gravity of this tick (systemDynamicsVSMPC.cpp:308)                   */
#define VSMPC_IN_GRAV 40      /*  3 Robot::getGravity (systemDynamicsVSMPC.cpp:309)                            */
#define VSMPC_IN_AMOM 43      /* 24 Robot::getMatrixAmomJets(true), 6x4 row-major (:93,:304)                   */
#define VSMPC_IN_LLIN 67      /* 24 Lambda_lin,B 3x8 row-major (systemDynamicsVSMPC.cpp:321-350)               */
#define VSMPC_IN_LANG 91      /* 24 Lambda_ang,B 3x8 row-major (systemDynamicsVSMPC.cpp:159-206)               */
#define VSMPC_IN_INERTIA 115  /*  9 I_G 3x3 row-major (systemDynamicsVSMPC.cpp:128-130)                        */
#define VSMPC_IN_RPY 124      /*  3 base RPY used for W^-1 (systemDynamicsVSMPC.cpp:132-147)                   */
#define VSMPC_IN_PREF 127     /*  3 QPInput::getPosCoMReference (systemDynamicsVSMPC.cpp:316)                  */
#define VSMPC_IN_RPYINIT 130  /*  3 configure-time RPY m_rpyInit (systemDynamicsVSMPC.cpp:67,100)              */
#define VSMPC_IN_T0 133       /*  4 linearisation thrust (systemDynamicsVSMPC.cpp:401-409)                     */
#define VSMPC_IN_TD0 137      /*  4 linearisation thrust rate                                                  */
#define VSMPC_IN_UPREV 141    /*  4 QPInput::getThrottleMPC, percent (systemDynamicsVSMPC.cpp:411)             */
#define VSMPC_IN_TDES 145     /*  4 QPInput::getThrustDesMPC (systemDynamicsVSMPC.cpp:415)                     */
#define VSMPC_IN_TDDES 149    /*  4 QPInput::getThrustDotDesMPC                                                */
#define VSMPC_IN_QERR 153     /*  8 q_cmd,sel - q_ref0 (costsVSMPC.cpp:574-589)                                */
#define VSMPC_IN_HOLD 161     /*  1 != 0 when the 20-tick hold pins v0 this tick (constraintsVSMPC.cpp:351)    */
#define VSMPC_IN_XREF 162     /* 12*(nIter-nIterSmall+1): reference window, column-major xref[col*12+row],
                                    rows = CoM, h_lin, RPY, h_ang (costsVSMPC.cpp:96-99,183-264)               */

/* First-move block written per instance (variableSamplingMPC.cpp:99-102,138-151), 24 doubles. */
#define VSMPC_FM_DQ 0          /* 8 joint-position increments  x[off_joints .. +8]             */
#define VSMPC_FM_V0 8          /* 4 warped throttle v0         x[off_throttle .. +4]           */
#define VSMPC_FM_THROTTLE 12   /* 4 throttle percent, JetModel::destandardizeThrottle_u2T(v0)  */
#define VSMPC_FM_THRUST 16     /* 4 thrust reference    = node 1 thrust  X1[12:16]             */
#define VSMPC_FM_THRUSTDOT 20  /* 4 thrust-rate reference = node 1      X1[16:20]              */
#define VSMPC_FM_SIZE 24

typedef struct vsmpc_handle vsmpc_handle;

/* Replaces IMPCProblem::configure (IMPCProblem.cpp:3-148) + VariableSamplingMPC::setCostAndConstraints
 * (variableSamplingMPC.cpp:7-86): validates the configuration, selects the kernel instantiation and
 * allocates every device and staging buffer for `max_batch` instances on HIP device `device` (solve, linearise,
 * kinematics, diagnostics).  No allocation happens in any later call on the handle; a rollout object allocates at its
 * own create and grows its log buffer only when a longer logged run is requested.
 * Horizons: any (nIter, nIterSmall, controlHorizon) listed in csrc/vsmpc_horizons.def has a kernel instantiation
 * (the kernels are straight-line code generated per horizon); others return VSMPC_ERR_UNSUPPORTED_CONFIG -- add the
 * horizon to VSMPC_HORIZONS and rebuild. */
int vsmpc_create(const vsmpc_config* cfg, int device, int max_batch, vsmpc_handle** out);
void vsmpc_destroy(vsmpc_handle* h);

/* IMPCProblem::getNOptimizationVariables / getNConstraints (IMPCProblem.h:71-78) and record sizes. */
int vsmpc_num_variables(const vsmpc_handle* h);    /* nVar  = 26(N+1)+8H+4(H-nS+1)  (588) */
int vsmpc_num_constraints(const vsmpc_handle* h);  /* nCon  = 26(N+1)+4(N-nS+1)     (512) */
int vsmpc_input_doubles(const vsmpc_handle* h);    /* n_in  = 162+12(N-nS+1)        (294) */
int vsmpc_max_batch(const vsmpc_handle* h);

/* Replaces update()+solveMPC() for `batch` independent instances (variable_sampling_mpc.py:111-112).
 * The caller's current HIP device is left as it was found (every entry point of this header does so).
 * Host buffers: in[batch*n_in]; x[batch*nVar] primal in the REFERENCE variable order
 * [X0..XN | U0..U_{H-1} | v0..v_{H-nS}]; first_move[batch*24]; status[batch]; iters[batch] (active-set
 * iterations).  x, first_move, iters may be NULL.  `stream` is a hipStream_t (NULL = default stream);
 * the call returns after the results are in the host buffers. */
int vsmpc_solve_batch(vsmpc_handle* h, const double* in, int batch, double* x, double* first_move,
                      int* status, int* iters, void* stream);

/* Same, all pointers are DEVICE pointers and the call only enqueues work on `stream` (no copies,
 * no synchronisation): the form a resident batch driver uses.  batch <= vsmpc_max_batch(h). */
int vsmpc_solve_batch_device(vsmpc_handle* h, const double* d_in, int batch, double* d_x,
                             double* d_first_move, int* d_status, int* d_iters, void* stream);

/* Parity split of the path: only the per-tick linearisation + discretisation, i.e.
 * SystemDynamicVS::updateDynamicMatrices/getAMatrix/getBJointsMatrix/getBThrottleMatrix/getCVector
 * (systemDynamicsVSMPC.cpp:495-585) and the dt schedule of constraintsVSMPC.cpp:45-51,78-84.
 * Host buffers, row-major per instance: A[batch*676], Bj[batch*208], Bt[batch*104], c[batch*26],
 * dt[nIter] (same for every instance). */
int vsmpc_linearize_batch(vsmpc_handle* h, const double* in, int batch, double* A, double* Bj,
                          double* Bt, double* c, double* dt);

/* Kinematics-derived inputs on the device (rows a3/a4 of the path): LinearMomentumDynamicVS::computeLambdaLin
 * (systemDynamicsVSMPC.cpp:321-350), AngularMomentumDynamicVS::computeLambdaAng + getRelativeJacobianCoM
 * (:159-226, "unfiltered" option) and the locked inertia of updateRPY (:128-130), from the raw Robot quantities.
 * kin[batch][VSMPC_KIN_SIZE] host buffer; out[batch][57] = Lambda_lin,B (24) | Lambda_ang,B (24) | I_G (9), all
 * row-major.  If `records` is not NULL (host, [batch][n_in]) the three fields are also written into the input
 * records at VSMPC_IN_LLIN / VSMPC_IN_LANG / VSMPC_IN_INERTIA. */
#define VSMPC_KIN_NJ 23        /* robot joints (MPCPyBindings.cpp:43) */
#define VSMPC_KIN_WRB 0        /*   9 wR_b row-major                                                   */
#define VSMPC_KIN_THRUST 9     /*   4 Robot::getJetThrusts                                             */
#define VSMPC_KIN_AXES 13      /*  12 Robot::getMatrixOfJetAxes 4x3                                    */
#define VSMPC_KIN_ARMS 25      /*  12 Robot::getMatrixOfJetArms 4x3                                    */
#define VSMPC_KIN_JREL 37      /* 276 getRelativeJacobianJetsBodyFrame()[i].bottomRows(3), 4 x (3x23)   */
#define VSMPC_KIN_JFRAME 313   /* 276 getJacobian(jet).topRightCorner(3,23), 4 x (3x23)                 */
#define VSMPC_KIN_JCOM 589     /*  69 getJacobianCoM().topRightCorner(3,23)                            */
#define VSMPC_KIN_MB 658       /*  36 getMassMatrix().block(0,0,6,6) row-major                         */
#define VSMPC_KIN_R 694        /*   3 p_CoM - p_base                                                   */
#define VSMPC_KIN_SIZE 697
#define VSMPC_KIN_OUT 57
int vsmpc_kinematics_batch(vsmpc_handle* h, const double* kin, int batch, double* out, double* records);

/* One tick of the reference's surface in ONE submission: update() [the kinematics-derived terms, systemDynamicsVSMPC.cpp:
 * 128-130,159-226,321-350] + solveMPC() [IMPCProblem.cpp:196-298, variableSamplingMPC.cpp:88-112] as the harness issues
 * them back to back (src/variable_sampling_mpc.py:111-112).  kin[batch][VSMPC_KIN_SIZE] and in[batch][n_in] are host
 * buffers; the device writes Lambda_lin,B | Lambda_ang,B | I_G into the records, solves them, and the call returns after
 * ONE synchronisation with the completed fields copied back into `in` (so `in` is the record that was solved) and the
 * results in x / first_move / status / iters (x, first_move, iters may be NULL).  batch <= 8 runs through the handle's
 * pinned, device-mapped staging buffer without any copy engine work: the form a 200 Hz single-robot caller uses
 * (include/VariableSamplingMPC.hpp, TickMachineT).  No allocation. */
int vsmpc_tick(vsmpc_handle* h, const double* kin, double* in, int batch, double* x, double* first_move, int* status,
               int* iters, void* stream);

/* Options of vsmpc_kinematics_batch, per handle.
 *   joint_selector[8] (or NULL = keep): robot joint index of every controlled joint, for the columns of Lambda_ang -- the
 *       reference selects them by NAME (`controlledJoints` against Robot::getJointName, systemDynamicsVSMPC.cpp:57-66,
 *       202-205).  Default 3..10, the shipped robot.  Lambda_lin keeps the reference's hard-coded offset 3 (:348).
 *   constant_lambda: jointsLambdaOption "constant" (systemDynamicsVSMPC.cpp:186-200,329-337).  The caller then delivers
 *       the CONFIGURE-TIME axes, arms and relative Jacobians; the JFRAME slot holds the relative Jacobians' TOP rows
 *       (4 x (3x23)) and JCOM[0..3] the thrusts of getRobot() that scale the angular term (the linear term keeps
 *       THRUST = getRobotReference()'s, as in the reference). */
int vsmpc_set_kinematics_options(vsmpc_handle* h, const int* joint_selector, int constant_lambda);

/* Batched kinematics provider: what the reference's Robot::setState caches for the path
 * (utils/src/Robot.cpp:198-335, getJacobian :505-514) on a SIMPLIFIED tree handed over as plain arrays: a floating base,
 * 8 revolute joints (joint j moves body j + 1; parents precede children), 4 jet frames.  iDynTree's conventions: MIXED
 * velocity representation, free-floating Jacobians [linear; angular] x [base 6 | joints].  The reference uses iDynTree
 * on the iRonCub URDF, neither of which is in this image: parity of this row is unpinned (oracle/robot_tree_ref.py). */
#define VSMPC_TREE_NB 9
#define VSMPC_TREE_NJ 8
typedef struct {
    int parent[VSMPC_TREE_NB];                 /* parent body (parent[0] = -1: the floating base) */
    int robot_joint[VSMPC_TREE_NJ];            /* index of tree joint j among the robot's 23 joints (Jacobian column) */
    double joint_axis[3 * VSMPC_TREE_NJ];      /* in the parent body's frame */
    double joint_origin[3 * VSMPC_TREE_NJ];    /* in the parent body's frame; the child frame sits there */
    double mass[VSMPC_TREE_NB];
    double com[3 * VSMPC_TREE_NB];             /* body frame */
    double inertia[6 * VSMPC_TREE_NB];         /* about the body CoM, body axes: xx xy xz yy yz zz */
    int jet_body[VSMPC_N_THRUSTS];
    double jet_origin[3 * VSMPC_N_THRUSTS];    /* body frame */
    double jet_axis[3 * VSMPC_N_THRUSTS];      /* thrust force direction, body frame (m_jetsAxesLocalFrames, Robot.cpp:256) */
    double gravity[3];
} vsmpc_tree;
/* state record per instance */
#define VSMPC_RS_P 0      /*  3 base position (world)                         */
#define VSMPC_RS_R 3      /*  9 wR_b row-major                                 */
#define VSMPC_RS_V 12     /*  3 base linear velocity (world, MIXED)           */
#define VSMPC_RS_W 15     /*  3 base angular velocity (world)                 */
#define VSMPC_RS_Q 18     /*  8 joint positions                               */
#define VSMPC_RS_QD 26    /*  8 joint velocities                              */
#define VSMPC_RS_T 34     /*  4 jet thrusts                                   */
#define VSMPC_RS_SIZE 38
/* Robot-level outputs per instance */
#define VSMPC_RO_COM 0    /*  3 getPositionCoM                                 */
#define VSMPC_RO_MOM 3    /*  6 getMomentum(false): centroidal, world axes     */
#define VSMPC_RO_MOMB 9   /*  6 getMomentum(true)                              */
#define VSMPC_RO_MASS 15  /*  1 getTotalMass                                   */
#define VSMPC_RO_AMOM 16  /* 24 getMatrixAmomJets(false) 6x4 row-major         */
#define VSMPC_RO_AMOMB 40 /* 24 getMatrixAmomJets(true)                        */
#define VSMPC_RO_RPY 64   /*  3 getBasePose().getRotation().asRPY()            */
#define VSMPC_RO_SIZE 67
/* state[batch][VSMPC_RS_SIZE] -> kin[batch][VSMPC_KIN_SIZE] (the record vsmpc_kinematics_batch reads; NULL = not wanted),
 * robot[batch][VSMPC_RO_SIZE] (NULL = not wanted) and, when `records` (batch input records of this handle's size) is
 * given, the fields update() pulls out of the Robot patched in place: X0 position / momentum / RPY / thrusts, MASS, WRB,
 * OMEGA, GRAV, AMOM, RPY, T0, and -- through the kinematics kernel on the device, no host round trip -- LLIN, LANG,
 * INERTIA.  Everything else of the record (references, throttle feedback, hold flag) stays the caller's. */
/* With jointsLambdaOption "constant" set on the handle, patching `records` is refused (VSMPC_ERR_UNSUPPORTED_CONFIG): that
 * option re-reads the Jacobian slots as configure-time quantities the provider does not deliver. */
int vsmpc_provider_batch(vsmpc_handle* h, const vsmpc_tree* tree, const double* state, int batch, double* kin,
                         double* robot, double* records);

/* Debug/parity: the reference-ordered dense QP of ONE instance, assembled on the host from the
 * DEVICE linearisation exactly as IMPCProblem::update stacks it (IMPCProblem.cpp:150-194):
 * H[nVar*nVar], g[nVar], Ac[nCon*nVar] (row-major), lo[nCon], hi[nCon]. */
int vsmpc_assemble_dense(vsmpc_handle* h, const double* in_one, double* H, double* g, double* Ac,
                         double* lo, double* hi);

/* Debug/parity: condensed problem of ONE instance after the device condensing + factorisation
 * phases.  M[np*np] row-major lower triangle of the augmented condensed Hessian (np =
 * vsmpc_condensed_dim(h); row NZ holds the condensed gradient), L likewise after the Cholesky. */
int vsmpc_condensed_dim(const vsmpc_handle* h);
int vsmpc_debug_condensed(vsmpc_handle* h, const double* in_one, double* M, double* Lfac);

/* Diagnostic build of the solve kernel (separate instantiation, never the timed one): s_memtime stamps of
 * workgroup thread 0 at the 10 phase boundaries P0..P6 of every instance, stamps16[batch*16]. */
int vsmpc_debug_phase_cycles(vsmpc_handle* h, const double* in, int batch, unsigned long long* stamps16);

/* Average device time per launch of the solve kernel over the launches enqueued between
 * vsmpc_timing_begin and vsmpc_timing_end on the handle's stream, measured with HIP events on the
 * stream the kernel is launched on.  Returns milliseconds through *ms_per_launch. */
int vsmpc_timing_begin(vsmpc_handle* h, void* stream);
int vsmpc_timing_end(vsmpc_handle* h, void* stream, int launches, float* ms_per_launch);

/*
 * ---- Closed-loop batched rollout --------------------------------------------------------------
 * The per-tick "advance" the reference hides inside its plugins and harness, for `batch` instances resident in HBM:
 *   build record (tick state machine: 20-tick hold constraintsVSMPC.cpp:351-372, reference window costsVSMPC.cpp:
 *   124-165, alpha cursor systemDynamicsVSMPC.cpp:308-311)  ->  solve  ->  apply the first move only if Solved
 *   (variableSamplingMPC.cpp:91-108)  ->  integrate a centroidal + jet plant over periodMPC in 1 ms sub-steps
 *   (the harness steps its simulator 5 x 1 ms per tick, ironcub_mujoco_simulator.py:122-139).
 * The plant is a synthetic stand-in for MuJoCo (out of scope): the centroidal-momentum model the MPC linearises with
 * the non-linear rotation kinematics and the non-linear jet polynomial (utils/src/JetModel.cpp:29-64).
 *
 * Plant state per instance, VSMPC_PLANT_STATE doubles: */
#define VSMPC_PS_P 0        /* 3 CoM position                                   */
#define VSMPC_PS_HLIN 3     /* 3 linear momentum, body frame                    */
#define VSMPC_PS_RPY 6      /* 3 base roll/pitch/yaw                            */
#define VSMPC_PS_HANG 9     /* 3 angular momentum, body frame                   */
#define VSMPC_PS_T 12       /* 4 jet thrusts                                    */
#define VSMPC_PS_TD 16      /* 4 jet thrust rates                               */
#define VSMPC_PS_Q 20       /* 8 controlled joint positions                     */
#define VSMPC_PS_U 28       /* 4 throttle command in percent (held between MPC moves) */
#define VSMPC_PS_TDES 32    /* 4 last MPC thrust reference   (QPInput::getThrustDesMPC)    */
#define VSMPC_PS_TDDES 36   /* 4 last MPC thrust-rate reference                            */
/* jet plant option (vsmpc_rollout_set_jet_plant, include/vsmpc_jet.h): NN thrust dynamics + EKF estimates; unused and
 * untouched with the polynomial jet plant */
#define VSMPC_PS_TNN 40     /* 4 thrust of the LSTM jet plant (float32 values), fed back to it     */
#define VSMPC_PS_EST 44     /* 8 EKF estimate (T, Tdot) per jet: what the MPC sees as measurement  */
#define VSMPC_PS_EKFP 52    /* 16 EKF covariance per jet, 2x2 row-major                           */
#define VSMPC_PLANT_STATE 68
/* Plant parameters per instance (constant over a rollout), VSMPC_PLANT_PARAMS doubles: */
#define VSMPC_PP_MASS 0
#define VSMPC_PP_INERTIA_B 1   /*   9 body-frame locked inertia, row-major                                          */
#define VSMPC_PP_AMOM0 10      /*  24 A_mom(q_ref0) 6x4 row-major, body frame                                      */
#define VSMPC_PP_DJ 34         /* 192 dA_mom/dq_j, [8][6][4]: A_mom(q) = A_mom0 + sum_j DJ[j] (q_j - q_ref0_j);
                                      Lambda_lin/ang column j = DJ[j] T (systemDynamicsVSMPC.cpp:159-206,321-350)   */
#define VSMPC_PP_QREF0 226     /*   8 posture reference q_ref0 (costsVSMPC.cpp:574-589)                             */
#define VSMPC_PP_PINIT 234     /*   3 configure-time CoM position (costsVSMPC.cpp:105)                              */
#define VSMPC_PP_RPYINIT 237   /*   3 configure-time RPY                                                            */
#define VSMPC_PP_DIST_F 240    /*   3 disturbance force, world frame [N]                                            */
#define VSMPC_PP_DIST_TAU 243  /*   3 disturbance torque, body frame [Nm]                                           */
#define VSMPC_PP_DIST_T0 246   /*   disturbance active for DIST_T0 <= t < DIST_T1 [s]                               */
#define VSMPC_PP_DIST_T1 247
#define VSMPC_PP_TICK0 248     /*   tick index of this instance at rollout start (phase of the hold / trajectory)   */
#define VSMPC_PLANT_PARAMS 249
/* Log row per tick and instance: p(3) rpy(3) T(4) throttle%(4) status iters */
#define VSMPC_ROLLOUT_LOG 16

typedef struct vsmpc_rollout vsmpc_rollout;

/* Allocates the resident rollout state for `batch` <= vsmpc_max_batch(h) instances.  The trajectories are shared by
 * all instances, as in the reference's TrajectoryManager (TrajectoryManager.cpp:23-39): CoM position OFFSETS and
 * velocities sampled every periodMPCLargeSteps (`n_traj` samples, [n_traj][3]), alpha_gravity sampled every
 * `alpha_dt` seconds (`n_alpha` samples, linearly up-sampled to the tick rate); both are clamped at their ends. */
int vsmpc_rollout_create(vsmpc_handle* h, int batch, const double* traj_pos, const double* traj_vel, int n_traj,
                         const double* traj_alpha, int n_alpha, double alpha_dt, vsmpc_rollout** out);
void vsmpc_rollout_destroy(vsmpc_rollout* r);
/* Host -> device: state[batch][VSMPC_PLANT_STATE], params[batch][VSMPC_PLANT_PARAMS]; resets the tick counters and
 * builds the record of tick 0 on the device. */
int vsmpc_rollout_reset(vsmpc_rollout* r, const double* state, const double* params);
/* Runs `ticks` closed-loop ticks on `stream`; log (host, may be NULL) receives [ticks][batch][VSMPC_ROLLOUT_LOG].
 * Returns after the last tick has completed. */
int vsmpc_rollout_run(vsmpc_rollout* r, int ticks, double* log, void* stream);
/* Device -> host copies of the current plant state / of the records the NEXT tick will solve ([batch][n_in], parity
 * hook: after reset the records of tick 0, after run(k) those of tick k). */
/* RPY / RPYDot tracks of the position trajectory (costsVSMPC.cpp:110-112,141-146: RPY reference = configure-time RPY +
 * track, angular-momentum reference = I_G W RPYDot at the attitude of the push), [n_traj][3] each, at the position
 * trajectory's rate; NULL = all zero (the shipped files).  Call before vsmpc_rollout_reset. */
int vsmpc_rollout_set_attitude_tracks(vsmpc_rollout* r, const double* traj_rpy, const double* traj_rpy_dot);
/* Kinematic-tree plant: with a tree set, the plant's joint vectoring IS the kinematics the MPC linearises -- every tick the
 * batched kinematics provider (vsmpc_provider_batch's device code) is evaluated on the plant's own joint state, in the body
 * frame, and A_mom,body(q), the locked inertia I_B(q) and Lambda_lin,B / Lambda_ang,B (through vsmpc_kinematics_batch's
 * device code, from the tree's Jacobians) replace the plant parameters AMOM0 / DJ / INERTIA_B: what the harness does when it
 * refreshes the Robot from the simulator state every tick (src/mujoco_lib/ironcub_mujoco_simulator.py:318-346 ->
 * utils/src/Robot.cpp:198-335).  PP_MASS must be the tree's total mass; the controlled joints are the tree's eight
 * (tree->robot_joint[j] = 3 + j for the default selector).  NULL switches back to the parametric plant.  Call before
 * vsmpc_rollout_reset; a tick is then six launches instead of two (still replayed from a captured graph). */
int vsmpc_rollout_set_tree(vsmpc_rollout* r, const vsmpc_tree* tree);
int vsmpc_rollout_get_state(vsmpc_rollout* r, double* state);
int vsmpc_rollout_get_records(vsmpc_rollout* r, double* records);

/* Pinned host memory for the caller's in/x/first_move/status buffers (thin wrappers of hipHostMalloc / hipHostFree,
 * so that a C caller need not link the HIP runtime).  With pinned OUTPUT buffers vsmpc_solve_batch lets the kernel write
 * the results straight into them (no device-to-host copies) while the uploads of the next chunk overlap the solves of
 * the current one; with pageable buffers the runtime stages the copies itself.  Returns NULL on failure. */
void* vsmpc_alloc_host(size_t bytes);
void vsmpc_free_host(void* p);

/* How the solve kernel condenses the QP (what constraintsVSMPC.cpp:76-131 / costsVSMPC.cpp:166-200 imply once the states
 * are eliminated).  STRUCTURED (form 1; the default where the horizon has it: nIter <= 18, even controlHorizon): forward /
 * adjoint recursions on 3 generator columns per joint block + the throttle columns, O(N^2) small 3x3 work.  SYRK (form 2;
 * every horizon): the sensitivity recursion of all condensed columns + C = sum_k Y_k^T Y_k on the matrix cores.  Form 0
 * restores the horizon's default.  The two forms agree to rounding (different summation order), not bit for bit.
 * Per handle; a new handle starts with the default (or with VSMPC_FORM=structured|syrk from the environment).
 * Returns the previous setting, VSMPC_ERR_INVALID_ARG, or VSMPC_ERR_UNSUPPORTED_CONFIG (form 1 without the instantiation). */
int vsmpc_set_kernel_form(vsmpc_handle* h, int form);

const char* vsmpc_strerror(int code);
const char* vsmpc_kernel_name(const vsmpc_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* VSMPC_H */
