"""Host-to-host rate of vsmpc_solve_batch (PCIe-inclusive): pageable numpy buffers vs pinned buffers
(vsmpc_alloc_host), batch 256 and 4096.  python tools/hostpath.py"""
import ctypes, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"
L = importlib.import_module(PKG + ".layout"); S = importlib.import_module(PKG + ".synth"); solver = importlib.import_module(PKG + ".solver")
_lib = importlib.import_module(PKG + "._lib")
lib = _lib.load()
cfg = L.paper_config()
out = {}
for B in (256, 4096):
    recs = S.make_batch(cfg, B, workload="montecarlo")
    m = solver.BatchedVSMPC(cfg, device=0, max_batch=B)
    for kind in ("pageable", "pinned"):
        if kind == "pinned":
            def pinned(shape, dtype):
                n = int(np.prod(shape)) * np.dtype(dtype).itemsize
                p = lib.vsmpc_alloc_host(n)
                return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), (n,)).view(dtype).reshape(shape), p
            inp, p0 = pinned((B, cfg.n_in), np.float64); inp[:] = recs
            x, p1 = pinned((B, cfg.n_var), np.float64); fm, p2 = pinned((B, 24), np.float64)
            st, p3 = pinned((B,), np.int32); it, p4 = pinned((B,), np.int32)
        else:
            inp = recs.copy(); x = np.empty((B, cfg.n_var)); fm = np.empty((B, 24)); st = np.empty(B, np.int32); it = np.empty(B, np.int32)
        ptr = lambda a: ctypes.c_void_p(a.ctypes.data)    # taken once: building the ctypes view of a large array per call costs ms
        p_in, p_x, p_fm, p_st, p_it = ptr(inp), ptr(x), ptr(fm), ptr(st), ptr(it)
        call = lambda: _lib.check(lib.vsmpc_solve_batch(m._h, p_in, B, p_x, p_fm, p_st, p_it, None))
        for _ in range(5): call()
        t = time.perf_counter(); n = 30
        for _ in range(n): call()
        dt = (time.perf_counter() - t) / n
        assert (st == 1).all()
        out[f"{kind}:{B}"] = {"us_per_call": dt * 1e6, "solves_per_s": B / dt}
        # first-move-only output (x = NULL): the harness' own use
        call2 = lambda: _lib.check(lib.vsmpc_solve_batch(m._h, p_in, B, None, p_fm, p_st, p_it, None))
        for _ in range(3): call2()
        t = time.perf_counter()
        for _ in range(n): call2()
        dt = (time.perf_counter() - t) / n
        out[f"{kind}:{B}:first_move_only"] = {"us_per_call": dt * 1e6, "solves_per_s": B / dt}
    m.close()
print(json.dumps(out, indent=1))
