"""ctypes front end of oracle/_build/liboracle.so (the C restatement: reference-ordered dense assembly + OSQP-style
ADMM/LDL'/polish).  ORACLE — test infrastructure only; parity unpinned (see vsmpc_oracle.c)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liboracle.so")


class Cfg(ctypes.Structure):
    _fields_ = [("n_iter", ctypes.c_int), ("n_iter_small", ctypes.c_int), ("control_horizon", ctypes.c_int),
                ("use_jet_dynamic", ctypes.c_int), ("period_mpc", ctypes.c_double), ("period_small", ctypes.c_double),
                ("period_large", ctypes.c_double), ("w_com_pos", ctypes.c_double * 3),
                ("w_com_pos_err", ctypes.c_double * 3), ("w_lin_mom", ctypes.c_double * 3),
                ("w_rpy", ctypes.c_double * 3), ("w_rpy_err", ctypes.c_double * 3), ("w_ang_mom", ctypes.c_double * 3),
                ("w_delta_joint", ctypes.c_double * 8), ("w_throttle", ctypes.c_double),
                ("w_initial_throttle", ctypes.c_double), ("w_reg_joint_pos", ctypes.c_double),
                ("throttle_min", ctypes.c_double), ("throttle_max", ctypes.c_double)]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or any(os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, f))
                                          for f in ("vsmpc_oracle.c", "vsmpc_structured.c")):
            subprocess.run(["make", "-s", "-C", HERE], check=True)
        lib = ctypes.CDLL(LIB)
        vp, ip = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)
        lib.vso_workspace_create.restype = vp
        lib.vso_workspace_create.argtypes = [ctypes.POINTER(Cfg)]
        lib.vso_workspace_free.argtypes = [vp]
        lib.vso_solve.argtypes = [vp, vp, vp, vp, ip, ip, vp]
        lib.vso_solve.restype = ctypes.c_int
        lib.vso_assemble_dense.argtypes = [ctypes.POINTER(Cfg), vp, vp, vp, vp, vp, vp]
        lib.vso_linearize.argtypes = [ctypes.POINTER(Cfg), vp, vp, vp, vp, vp]
        lib.vso_dt_schedule.argtypes = [ctypes.POINTER(Cfg), vp]
        lib.vso_time_batch.argtypes = [ctypes.POINTER(Cfg), vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ip, vp, vp]
        lib.vso_time_batch.restype = ctypes.c_double
        lib.vso_max_threads.restype = ctypes.c_int
        lib.vso_time_batch2.argtypes = [ctypes.POINTER(Cfg), vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ip, vp, vp, ctypes.c_int]
        lib.vso_time_batch2.restype = ctypes.c_double
        lib.vss_solve.argtypes = [ctypes.POINTER(Cfg), vp, vp, ip]
        lib.vss_solve.restype = ctypes.c_int
        lib.vss_time_batch.argtypes = [ctypes.POINTER(Cfg), vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ip, vp, vp]
        lib.vss_time_batch.restype = ctypes.c_double
        _lib = lib
    return _lib


def make_cfg(rcfg) -> Cfg:
    """rcfg: oracle.vsmpc_ref.Config (same keys as VS_MPC_CONFIG)."""
    c = Cfg()
    c.n_iter, c.n_iter_small, c.control_horizon = rcfg.n_iter, rcfg.n_iter_small, rcfg.control_horizon
    c.use_jet_dynamic = 1 if rcfg.use_jet_dynamic else 0
    c.period_mpc, c.period_small, c.period_large = rcfg.period_mpc, rcfg.period_small, rcfg.period_large
    for k in ("w_com_pos", "w_com_pos_err", "w_lin_mom", "w_rpy", "w_rpy_err", "w_ang_mom"):
        setattr(c, k, (ctypes.c_double * 3)(*getattr(rcfg, k)))
    c.w_delta_joint = (ctypes.c_double * 8)(*rcfg.w_delta_joint)
    c.w_throttle, c.w_initial_throttle, c.w_reg_joint_pos = rcfg.w_throttle, rcfg.w_initial_throttle, rcfg.w_reg_joint_pos
    c.throttle_min, c.throttle_max = rcfg.throttle_min, rcfg.throttle_max
    return c


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def assemble_dense(rcfg, rec):
    lib, c = load(), make_cfg(rcfg)
    n, m = rcfg.n_var, rcfg.n_con
    rec = np.ascontiguousarray(rec, dtype=np.float64)
    H, g, Ac, lo, hi = np.empty((n, n)), np.empty(n), np.empty((m, n)), np.empty(m), np.empty(m)
    lib.vso_assemble_dense(ctypes.byref(c), _p(rec), _p(H), _p(g), _p(Ac), _p(lo), _p(hi))
    return H, g, Ac, lo, hi


def linearize(rcfg, rec):
    lib, c = load(), make_cfg(rcfg)
    rec = np.ascontiguousarray(rec, dtype=np.float64)
    A, Bj, Bt, cv, dt = np.empty((26, 26)), np.empty((26, 8)), np.empty((26, 4)), np.empty(26), np.empty(rcfg.n_iter)
    lib.vso_linearize(ctypes.byref(c), _p(rec), _p(A), _p(Bj), _p(Bt), _p(cv))
    lib.vso_dt_schedule(ctypes.byref(c), _p(dt))
    return A, Bj, Bt, cv, dt


class Solver:
    """update()+solve() of one instance at a time with the OSQP-style restatement (symbolic set-up once)."""

    def __init__(self, rcfg):
        self.lib, self.rcfg, self.c = load(), rcfg, make_cfg(rcfg)
        self.ws = self.lib.vso_workspace_create(ctypes.byref(self.c))

    def solve(self, rec):
        rec = np.ascontiguousarray(rec, dtype=np.float64)
        x, y = np.empty(self.rcfg.n_var), np.empty(self.rcfg.n_con)
        it, pol = ctypes.c_int(0), ctypes.c_int(0)
        res = np.zeros(2)
        st = self.lib.vso_solve(self.ws, _p(rec), _p(x), _p(y), ctypes.byref(it), ctypes.byref(pol), _p(res))
        return x, y, {"status": st, "iters": it.value, "polished": bool(pol.value), "prim_res": res[0], "dual_res": res[1]}

    def close(self):
        if self.ws:
            self.lib.vso_workspace_free(self.ws)
            self.ws = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def time_batch(rcfg, inputs, threads=None, budget_s=15.0, warm=False):
    lib, c = load(), make_cfg(rcfg)
    inputs = np.ascontiguousarray(inputs, dtype=np.float64)
    threads = threads or lib.vso_max_threads()
    done = ctypes.c_int(0)
    stats = np.zeros(3)
    x = np.full((inputs.shape[0], rcfg.n_var), np.nan)
    el = lib.vso_time_batch2(ctypes.byref(c), _p(inputs), inputs.shape[0], threads, budget_s, ctypes.byref(done), _p(x), _p(stats),
                             1 if warm else 0)
    return {"elapsed_s": el, "done": done.value, "threads": threads, "mean_iters": stats[0], "polished_frac": stats[1],
            "solved_frac": stats[2], "x": x}


def solve_structured(rcfg, rec):
    """vss_solve (vsmpc_structured.c): the structure-exploiting exact solve on one host thread."""
    lib, c = load(), make_cfg(rcfg)
    rec = np.ascontiguousarray(rec, dtype=np.float64)
    x = np.empty(rcfg.n_var)
    it = ctypes.c_int(0)
    st = lib.vss_solve(ctypes.byref(c), _p(rec), _p(x), ctypes.byref(it))
    return x, st, it.value


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def available_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup's CPU quota where there is one (a GPU
    box hands one job a share of the host: 16 CPUs per GPU on this pool, whatever the affinity mask says)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def host_threads(lib, avail):
    """worker threads of the CPU legs: every core this process may use (north_star: "the node's own host cores, core
    count stated"); VSMPC_CPU_THREADS caps it (it used to default to 16)"""
    cap = int(os.environ.get("VSMPC_CPU_THREADS", "0"))
    n = min(lib.vso_max_threads(), avail)
    return max(1, min(n, cap) if cap > 0 else n)


def time_structured(cfg_name, inputs, budget_s=8.0):
    """bench.py's `cpu_structured` leg: the condensed exact solve (the kernel's algorithm) in C on the host cores."""
    import vsmpc_ref as ref
    rcfg = ref.paper_config() if cfg_name == "paper" else ref.horizon2x_config()
    lib, c = load(), make_cfg(rcfg)
    avail = available_cpus()
    cores = host_threads(lib, avail)

    def run(arr, threads, budget):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        done, stats = ctypes.c_int(0), np.zeros(3)
        el = lib.vss_time_batch(ctypes.byref(c), _p(arr), arr.shape[0], threads, budget, ctypes.byref(done), None, _p(stats))
        return el, done.value, stats
    el1, n1, _ = run(inputs[:min(len(inputs), 256)], 1, budget_s / 4)
    ms1 = 1e3 * el1 / max(1, n1)
    want = int(cores * (budget_s * 3 / 4) / (ms1 * 1e-3)) + cores
    reps = max(1, -(-want // len(inputs)))
    el, n, stats = run(np.tile(inputs, (reps, 1)), cores, budget_s * 3 / 4)
    return {"value": n / el, "unit": "solves/s", "cores": cores, "cores_available": avail, "host_logical_cpus": os.cpu_count(), "kind": "port", "cpu": cpu_model(),
            "single_thread_ms_per_solve": ms1,
            "sample": f"{n} solves on {cores} threads in {el:.1f} s + {n1} single-thread solves ({ms1:.3f} ms each); scalar C "
                      f"of the kernel's own algorithm (condense -> Cholesky -> box QP on the throttles, mean "
                      f"{stats[0]:.2f} active-set iterations, {100 * stats[2]:.0f}% solved): separates the algorithmic gain "
                      f"from the hardware gain (BASELINE.md 4.2)"}


def time_baseline(cfg_name, inputs, budget_s=15.0):
    """bench.py's cpu_baseline leg: a single-thread latency sample, then a throughput sample on the host cores this
    process may use (all of them; `cores_available` = what the affinity mask allows), about `budget_s` seconds in all."""
    import vsmpc_ref as ref
    rcfg = ref.paper_config() if cfg_name == "paper" else ref.horizon2x_config()
    lib = load()
    avail = available_cpus()
    cores = host_threads(lib, avail)
    one = time_batch(rcfg, inputs[:min(len(inputs), 256)], threads=1, budget_s=budget_s / 8)
    ms1 = 1e3 * one["elapsed_s"] / max(1, one["done"])
    onew = time_batch(rcfg, inputs[:min(len(inputs), 256)], threads=1, budget_s=budget_s / 8, warm=True)
    ms1w = 1e3 * onew["elapsed_s"] / max(1, onew["done"])
    want = int(cores * (budget_s * 4 / 5) / (ms1 * 1e-3)) + cores          # enough work for the remaining budget
    reps = max(1, -(-want // len(inputs)))
    allc = time_batch(rcfg, np.tile(inputs, (reps, 1)), threads=cores, budget_s=budget_s * 4 / 5)
    return {"value": allc["done"] / allc["elapsed_s"], "unit": "solves/s", "cores": cores, "cores_available": avail, "host_logical_cpus": os.cpu_count(), "kind": "port",
            "cpu": cpu_model(),
            "single_thread_ms_per_solve": ms1, "single_thread_ms_per_solve_warm_started": ms1w,
            "warm_start_mean_iters": onew["mean_iters"],
            "sample": f"{allc['done']} solves (the benchmark batch repeated) on {cores} threads in {allc['elapsed_s']:.1f} s "
                      f"+ {one['done']} single-thread solves ({ms1:.2f} ms each); C restatement of the reference algorithm: "
                      f"dense plugin-order assembly -> sparse KKT LDL' -> OSQP-style ADMM (mean {allc['mean_iters']:.0f} iters) "
                      f"-> polish ({100 * allc['polished_frac']:.0f}% accepted), cold start per instance; warm-started from the "
                      f"previous instance's (x, y, rho) as the reference is from the previous tick (IMPCProblem.cpp:140): "
                      f"{ms1w:.2f} ms single-thread, mean {onew['mean_iters']:.0f} iters; the reference's own figure is "
                      f"2.18 ms/solve warm-started (poster, hardware unstated)"}
