"""Host-side mirror of the reference's operator surface for the MPC path, on top of the C-ABI.

`BatchedVSMPC`            the batch driver (what bench.py and the parity tests call).
`VariableSamplingMPC`     one instance with the reference's method names
                          (momentum-based-linear-mpc-lib/bindings/python/MPCPyBindings.cpp:22-90,
                          include/variableSamplingMPC/variableSamplingMPC.h:15-41): configure / update /
                          solveMPC / get*Reference, including "consume the solution only if Solved"
                          (variableSamplingMPC.cpp:91) and the joint accumulator (:104-108).

All numerics run in libvsmpc.so (HIP).  There is no CPU path here.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from . import layout as L
from .jet_model import JetModel


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


KERNEL_FORM_AUTO, KERNEL_FORM_STRUCTURED, KERNEL_FORM_SYRK = 0, 1, 2


class BatchedVSMPC:
    """`max_batch` independent MPC instances on one GPU (one workgroup per instance)."""

    def __init__(self, cfg: L.MPCConfig | None = None, device: int = 0, max_batch: int = 256):
        self.cfg = cfg or L.paper_config()
        self.lib = _lib.load()
        self._ccfg = self.cfg.to_c()
        self._h = ctypes.c_void_p()
        _lib.check(self.lib.vsmpc_create(ctypes.byref(self._ccfg), device, max_batch, ctypes.byref(self._h)),
                   "vsmpc_create")
        self.device = device
        self.max_batch = max_batch
        self.n_var = self.lib.vsmpc_num_variables(self._h)
        self.n_con = self.lib.vsmpc_num_constraints(self._h)
        self.n_in = self.lib.vsmpc_input_doubles(self._h)
        self.n_p = self.lib.vsmpc_condensed_dim(self._h)
        assert self.n_var == self.cfg.n_var and self.n_in == self.cfg.n_in and self.n_con == self.cfg.n_con

    def set_kernel_form(self, form: int) -> int:
        """vsmpc_set_kernel_form (include/vsmpc.h): how this handle's solve kernel condenses the QP (structured
        forward / adjoint recursions, or sensitivity recursion + SYRK); returns the previous setting."""
        prev = self.lib.vsmpc_set_kernel_form(self._h, int(form))
        if prev < 0:
            raise ValueError(f"kernel form {form}: {self.lib.vsmpc_strerror(prev).decode()}")
        return prev

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.vsmpc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def kernel_name(self) -> str:
        return self.lib.vsmpc_kernel_name(self._h).decode()

    # ---- host-buffer entry (vsmpc_solve_batch)
    def solve(self, inputs: np.ndarray):
        inputs = np.ascontiguousarray(inputs, dtype=np.float64)
        if inputs.ndim != 2 or inputs.shape[1] != self.n_in:
            raise ValueError(f"inputs must be [batch, {self.n_in}]")
        B = inputs.shape[0]
        x = np.empty((B, self.n_var))
        fm = np.empty((B, L.FM_SIZE))
        status = np.empty(B, dtype=np.int32)
        iters = np.empty(B, dtype=np.int32)
        _lib.check(self.lib.vsmpc_solve_batch(self._h, _ptr(inputs), B, _ptr(x), _ptr(fm), _ptr(status),
                                              _ptr(iters), None), "vsmpc_solve_batch")
        return x, fm, status, iters

    # ---- device-resident entry (vsmpc_solve_batch_device); arguments are torch CUDA tensors
    def solve_device(self, d_in, d_x, d_fm, d_status, d_iters, stream=None):
        import torch
        assert d_in.is_cuda and d_in.dtype == torch.float64 and d_in.is_contiguous()
        B = d_in.shape[0]
        s = stream if stream is not None else torch.cuda.current_stream(d_in.device)
        _lib.check(self.lib.vsmpc_solve_batch_device(
            self._h, ctypes.c_void_p(d_in.data_ptr()), B,
            ctypes.c_void_p(d_x.data_ptr()) if d_x is not None else None,
            ctypes.c_void_p(d_fm.data_ptr()) if d_fm is not None else None,
            ctypes.c_void_p(d_status.data_ptr()),
            ctypes.c_void_p(d_iters.data_ptr()) if d_iters is not None else None,
            ctypes.c_void_p(s.cuda_stream)), "vsmpc_solve_batch_device")

    def timing_begin(self, stream):
        _lib.check(self.lib.vsmpc_timing_begin(self._h, ctypes.c_void_p(stream.cuda_stream)), "vsmpc_timing_begin")

    def timing_end(self, stream, launches: int) -> float:
        ms = ctypes.c_float(0.0)
        _lib.check(self.lib.vsmpc_timing_end(self._h, ctypes.c_void_p(stream.cuda_stream), launches,
                                             ctypes.byref(ms)), "vsmpc_timing_end")
        return float(ms.value)

    # ---- parity splits
    def linearize(self, inputs: np.ndarray):
        inputs = np.ascontiguousarray(inputs, dtype=np.float64)
        B = inputs.shape[0]
        A = np.empty((B, 26, 26))
        Bj = np.empty((B, 26, 8))
        Bt = np.empty((B, 26, 4))
        c = np.empty((B, 26))
        dt = np.empty(self.cfg.n_iter)
        _lib.check(self.lib.vsmpc_linearize_batch(self._h, _ptr(inputs), B, _ptr(A), _ptr(Bj), _ptr(Bt), _ptr(c),
                                                  _ptr(dt)), "vsmpc_linearize_batch")
        return A, Bj, Bt, c, dt

    def assemble_dense(self, one_input: np.ndarray):
        one_input = np.ascontiguousarray(one_input, dtype=np.float64).reshape(-1)
        H = np.empty((self.n_var, self.n_var))
        g = np.empty(self.n_var)
        Ac = np.empty((self.n_con, self.n_var))
        lo = np.empty(self.n_con)
        hi = np.empty(self.n_con)
        _lib.check(self.lib.vsmpc_assemble_dense(self._h, _ptr(one_input), _ptr(H), _ptr(g), _ptr(Ac), _ptr(lo),
                                                 _ptr(hi)), "vsmpc_assemble_dense")
        return H, g, Ac, lo, hi

    def debug_condensed(self, one_input: np.ndarray):
        one_input = np.ascontiguousarray(one_input, dtype=np.float64).reshape(-1)
        M = np.empty((self.n_p, self.n_p))
        Lf = np.empty((self.n_p, self.n_p))
        _lib.check(self.lib.vsmpc_debug_condensed(self._h, _ptr(one_input), _ptr(M), _ptr(Lf)),
                   "vsmpc_debug_condensed")
        return M, Lf


    def kinematics(self, kin: np.ndarray, records: np.ndarray | None = None):
        """Lambda_lin,B, Lambda_ang,B, I_G from raw Robot quantities (vsmpc_kinematics_batch); optionally patches
        them into `records` in place."""
        kin = np.ascontiguousarray(kin, dtype=np.float64)
        if kin.ndim != 2 or kin.shape[1] != L.KIN_SIZE:
            raise ValueError(f"kin must be [batch, {L.KIN_SIZE}]")
        out = np.empty((kin.shape[0], L.KIN_OUT))
        if records is not None:
            assert records.flags["C_CONTIGUOUS"] and records.shape == (kin.shape[0], self.n_in)
        _lib.check(self.lib.vsmpc_kinematics_batch(self._h, _ptr(kin), kin.shape[0], _ptr(out), _ptr(records)),
                   "vsmpc_kinematics_batch")
        return out[:, 0:24].reshape(-1, 3, 8), out[:, 24:48].reshape(-1, 3, 8), out[:, 48:57].reshape(-1, 3, 3)

    def tick(self, kin: np.ndarray, records: np.ndarray):
        """vsmpc_tick: kinematics terms -> records -> solve in one submission (one synchronisation).  `records` is completed
        in place (LLIN | LANG | INERTIA).  Returns (x, first_move, status, iters) like solve()."""
        kin = np.ascontiguousarray(kin, dtype=np.float64)
        n = kin.shape[0]
        if kin.ndim != 2 or kin.shape[1] != L.KIN_SIZE:
            raise ValueError(f"kin must be [batch, {L.KIN_SIZE}]")
        assert records.dtype == np.float64 and records.flags["C_CONTIGUOUS"] and records.shape == (n, self.n_in)
        x = np.empty((n, self.n_var))
        fm = np.empty((n, L.FM_SIZE))
        st = np.zeros(n, dtype=np.int32)
        it = np.zeros(n, dtype=np.int32)
        _lib.check(self.lib.vsmpc_tick(self._h, _ptr(kin), _ptr(records), n, _ptr(x), _ptr(fm), _ptr(st), _ptr(it), None),
                   "vsmpc_tick")
        return x, fm, st, it

    def provider(self, tree: dict, states: np.ndarray, records: np.ndarray | None = None):
        """vsmpc_provider_batch: the Robot quantities of the path on the simplified tree (robot_tree.py).  Returns
        (kin[batch, KIN_SIZE], robot[batch, RO_SIZE]); `records`, when given, get the Robot-derived fields patched in
        place (including Lambda_lin / Lambda_ang / I_G through the kinematics kernel, on the device)."""
        from . import robot_tree as RT
        states = np.ascontiguousarray(states, dtype=np.float64)
        if states.ndim != 2 or states.shape[1] != RT.RS_SIZE:
            raise ValueError(f"states must be [batch, {RT.RS_SIZE}]")
        n = states.shape[0]
        kin = np.empty((n, L.KIN_SIZE))
        robot = np.empty((n, RT.RO_SIZE))
        if records is not None:
            assert records.flags["C_CONTIGUOUS"] and records.shape == (n, self.n_in)
        ctree = RT.to_c(tree)
        _lib.check(self.lib.vsmpc_provider_batch(self._h, ctypes.byref(ctree), _ptr(states), n, _ptr(kin), _ptr(robot),
                                                 _ptr(records)), "vsmpc_provider_batch")
        return kin, robot

    def set_kinematics_options(self, joint_selector=None, constant_lambda: bool = False):
        """vsmpc_set_kinematics_options: robot joint indices of the controlled joints (Lambda_ang columns; the
        reference selects them by name) and jointsLambdaOption 'constant'."""
        sel = None
        if joint_selector is not None:
            sel = (ctypes.c_int * 8)(*[int(v) for v in joint_selector])
        _lib.check(self.lib.vsmpc_set_kinematics_options(self._h, sel, 1 if constant_lambda else 0),
                   "vsmpc_set_kinematics_options")

    def phase_cycles(self, inputs: np.ndarray) -> np.ndarray:
        """Diagnostic build only: per-instance s_memtime stamps at the phase boundaries, [batch, 16]."""
        inputs = np.ascontiguousarray(inputs, dtype=np.float64)
        st = np.zeros((inputs.shape[0], 16), dtype=np.uint64)
        _lib.check(self.lib.vsmpc_debug_phase_cycles(self._h, _ptr(inputs), inputs.shape[0], _ptr(st)),
                   "vsmpc_debug_phase_cycles")
        return st


class VariableSamplingMPC:
    """Single-instance wrapper with the reference's method names (MPCPyBindings.cpp:22-90).

    The reference's update(QPInput) pulls the per-tick quantities out of a live iDynTree-backed Robot;
    that kinematics provider is outside the path, so `update` takes the already
    extracted input record (layout.IN_*), which is exactly what the device path consumes.  This class carries the
    joint accumulator and "consume only if Solved"; the tick-state fields of the record (hold flag, unwrapped RPY,
    reference window, alpha cursor) are the caller's here -- `reference_api.VariableSamplingMPC` is the mirror that
    takes a QPInput and runs the reference's whole tick state machine itself.
    """

    N_ROBOT_JOINTS = 23  # MPCPyBindings.cpp:43 hard-codes 23 joints
    JOINT_OFFSET = 3     # controlled joints occupy robot joints 3..10 (systemDynamicsVSMPC.cpp:348)

    def __init__(self):
        self._solver = None
        self._record = None
        self._status = 0

    def configure(self, cfg: L.MPCConfig, initial_joint_positions=None, device: int = 0) -> bool:
        self._solver = BatchedVSMPC(cfg, device=device, max_batch=1)
        self.cfg = cfg
        q0 = np.zeros(self.N_ROBOT_JOINTS) if initial_joint_positions is None else np.asarray(initial_joint_positions, float)
        self._jointsPositionReference = q0.copy()   # variableSamplingMPC.cpp:59-60
        self._thrustReference = np.zeros(4)
        self._thrustDotReference = np.zeros(4)
        self._throttleReference = np.zeros(4)       # stored as warped v (variableSamplingMPC.cpp:100)
        # what getThrottleReference returns before the first Solved tick: destandardizeThrottle_u2T of the zero-initialised
        # warped throttle, as the reference's getter computes it from its stored member (variableSamplingMPC.cpp:138-151)
        self._throttlePercent = np.array([JetModel().destandardizeThrottle_u2T(0.0)] * 4)
        self._deltaJoints = np.zeros(8)
        self._QPSolution = np.zeros(cfg.n_var)
        self._finalState = np.zeros(26)
        return True

    def update(self, record: np.ndarray) -> bool:
        rec = np.asarray(record, dtype=np.float64).reshape(-1)
        if rec.size != self.cfg.n_in:
            return False
        self._record = rec
        return True

    def solveMPC(self) -> bool:
        x, fm, status, _ = self._solver.solve(self._record[None, :])
        self._status = int(status[0])
        if self._status == L.STATUS_SOLVED:          # variableSamplingMPC.cpp:91
            self._QPSolution = x[0]
            self._deltaJoints = fm[0, L.FM_DQ:L.FM_DQ + 8]
            self._throttleReference = fm[0, L.FM_V0:L.FM_V0 + 4]
            self._throttlePercent = fm[0, L.FM_THROTTLE:L.FM_THROTTLE + 4]
            self._thrustReference = fm[0, L.FM_THRUST:L.FM_THRUST + 4]
            self._thrustDotReference = fm[0, L.FM_THRUSTDOT:L.FM_THRUSTDOT + 4]
            self._finalState = x[0, 26 * self.cfg.n_iter:26 * (self.cfg.n_iter + 1)]
            sel = slice(self.JOINT_OFFSET, self.JOINT_OFFSET + 8)
            self._jointsPositionReference[sel] += self._deltaJoints   # variableSamplingMPC.cpp:104-108
        return True                                   # the reference returns true regardless (:111)

    def getQPProblemStatus(self) -> int:
        return self._status

    def getJointsReferencePosition(self):
        return self._jointsPositionReference.copy()

    def getThrottleReference(self):
        return np.array(self._throttlePercent, copy=True)

    def getThrustReference(self):
        return np.array(self._thrustReference, copy=True)

    def getThrustDotReference(self):
        return np.array(self._thrustDotReference, copy=True)

    def getMPCSolution(self):
        return self._QPSolution[self.cfg.off_joints:].copy()

    def getFinalCoMPosition(self):
        return self._finalState[0:3].copy()

    def getFinalLinMom(self):
        return self._finalState[3:6].copy()

    def getFinalRPY(self):
        return self._finalState[6:9].copy()

    def getFinalAngMom(self):
        return self._finalState[9:12].copy()

    def getNStatesMPC(self) -> float:
        return 26.0

    def getNInputMPC(self) -> float:
        return 12.0
