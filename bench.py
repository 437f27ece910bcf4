#!/usr/bin/env python3
"""Benchmark of the batched multi-rate MPC solve path (BASELINE.json metric: MPC solves/s).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One *step* = one pass of the hot path (linearise -> condense -> factor -> box QP -> simulate) over one
batch of synthetic instances that is already resident in HBM.  At N=1 the batch is BASELINE.json configs[1]
(batch=256 hover initial states, paper horizon/rates).  With N > 1 ranks the line carries BASELINE.json configs[3] --
the Monte-Carlo batch sharded over the GPUs, 4096 instances with 4x wider scatter per rank, every instance seeded by its
GLOBAL index (1234 + index: the batch does not depend on the number of ranks), no data-path collective (weak scaling) --
and the 256-per-GPU figure moves into `extra_configs`.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd"

# SURVEY.md 8(d) / BASELINE.md section 3: algorithmic FLOPs and HBM bytes per solve
F_ALG = {"paper": 3.086e6, "horizon2x": 6.17e6}
BYTES_ALG = {"paper": 7064, "horizon2x": 12488}
FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector/matrix peak (AMD spec; not listed in MI355X_MICROARCH.md)


def cpu_structured(cfg_name: str, inputs: np.ndarray, budget_s: float = 8.0):
    """Second CPU line: the kernel's OWN algorithm (condense -> Cholesky -> box QP) as scalar C on the
    host cores, so that the algorithmic gain over the reference's OSQP route and the hardware gain can be told apart."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import oracle_c
        return oracle_c.time_structured(cfg_name, inputs, budget_s)
    except Exception as exc:  # pragma: no cover
        return {"value": None, "unit": "solves/s", "note": f"unavailable ({type(exc).__name__}: {exc})"}


def cpu_baseline(cfg_name: str, inputs: np.ndarray, budget_s: float = 15.0):
    """Times the oracle on this host's cores on a bounded sample of the same workload (rank 0, N=1 only).
    The oracle is test infrastructure: it is the thing timed here, never the thing shipped."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import oracle_c  # C restatement of the reference algorithm (assembly + OSQP-style ADMM + polish)
        return oracle_c.time_baseline(cfg_name, inputs, budget_s)
    except Exception as exc:  # pragma: no cover - the C oracle is built by __graft_entry__.build()
        note = f"C oracle unavailable ({type(exc).__name__}: {exc}); numpy restatement timed instead"
    import vsmpc_ref as ref
    rcfg = ref.paper_config() if cfg_name == "paper" else ref.horizon2x_config()
    n, t0 = 0, time.perf_counter()
    while n < len(inputs) and (time.perf_counter() - t0) < budget_s:
        ref.solve_instance(rcfg, inputs[n])
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "solves/s", "cores": 1, "kind": "port",
            "sample": f"{n} instances of the benchmark batch, numpy oracle (dense assembly + null-space "
                      f"active set), {dt:.1f} s; {note}"}


def kernel_form(mpc, batch, dev):
    """How the solve kernel of this handle condenses the QP (include/vsmpc.h, vsmpc_set_kernel_form): asked of the
    library, not guessed -- a handle accepts form 1 only if its horizon has the structured form."""
    from importlib import import_module
    solver = import_module("paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd.solver")
    try:
        prev = mpc.set_kernel_form(solver.KERNEL_FORM_STRUCTURED)
        mpc.set_kernel_form(prev)
        structured = prev != solver.KERNEL_FORM_SYRK
    except ValueError:
        structured = False
    wgs = "two workgroups per CU" if mpc.n_p <= 128 else "one workgroup per CU"
    return (("structured condensing (forward / adjoint recursions)" if structured else "sensitivity recursion + SYRK on the matrix cores")
            + f"; 4 wavefronts, {wgs}")


def counters_of(config: str, workload: str, batch: int):
    """Counter-derived figures of this configuration from the committed rocprofv3 --pmc passes (profiles/hbm_traffic.json,
    profiles/mfma_counters.json; tools/prof_r04.sh).  They describe the kernel version named in `counters_from`, measured
    on the builder's lease -- NOT this run.
      executed_flops_per_solve  what the wavefronts actually executed: FP64 matrix-core FLOPs (SQ_INSTS_VALU_MFMA_MOPS_F64
                                x 512) + FP64 vector FLOPs (64 lanes x (ADD + MUL + 2 FMA + TRANS) instructions, an upper
                                bound: masked-off lanes are counted).  `frac` is NOT built from it: SURVEY 8(d) fixes the
                                numerator to the method-independent F_alg.
      bound                     what the counters say the kernel waits for (summarize_mfma.py: matrix pipe busy share,
                                wave time parked / issuing)."""
    key = f"{config}:{workload}:{batch}"
    out = {"traffic": None, "mfma_busy_frac": None, "counters_from": None, "executed_flops_per_solve": None, "bound": None,
           "wave_time_shares": None}
    for fname, fields in (("hbm_traffic.json", (("bytes_per_launch", "traffic"),)),
                          ("mfma_counters.json", (("mfma_busy_frac", "mfma_busy_frac"), ("executed_flops_per_instance", "executed_flops_per_solve"),
                                                  ("bound", "bound"), ("wave_time_shares", "wave_time_shares")))):
        path = os.path.join(ROOT, "profiles", fname)
        if os.path.exists(path):
            rec = json.load(open(path)).get(key)
            if rec:
                for field, dst in fields:
                    out[dst] = rec.get(field)
                out["counters_from"] = rec.get("tag", out["counters_from"])
    return out


def surface_latency(ticks: int = 2000):
    """p50 / p99 of update(qpInput) + solveMPC() through the reference-side binding -- VariableSamplingMPCT<QPInput,
    TrajectoryManager> exactly as INTEGRATION.md section 2 shows it (tests/cpp/integration_snippet.cpp), over the
    signature stand-ins of tests/cpp/refstub, batch 1, one vsmpc_tick submission per tick -- measured by a compiled C++
    driver in its own process (std::chrono around the two calls).  What a 200 Hz drop-in user of the class sees."""
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import fake_provider as fp
        consts = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_constants.json")))
        traj = dict(np.load(os.path.join(ROOT, "tests", "golden", "reference_trajectories.npz")))
        tmp = tempfile.mkdtemp(prefix="vsmpc_surface_")
        exe = os.path.join(tmp, "reference_surface_driver")
        cpp = os.path.join(ROOT, "tests", "cpp")
        pkg_dir = os.path.join(ROOT, PKG)
        cmd = ["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(cpp, "refstub"), "-I", cpp,
               os.path.join(cpp, "reference_surface_driver.cpp"), "-o", exe, "-L", pkg_dir, "-lvsmpc", f"-Wl,-rpath,{pkg_dir}",
               "-Wl,-rpath,/opt/rocm/lib"]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            return {"note": "driver did not build: " + res.stderr[-300:]}
        sc = fp.Scenario(n_ticks=45, seed=17)
        scen = os.path.join(tmp, "scenario.bin")
        open(scen, "wb").write(sc.serialise(consts, traj, list(range(3, 11))).tobytes())
        out = {}
        for form, extra in (("fused", []), ("two_call", ["two-call"])):
            res = subprocess.run([exe, scen, "-", "latency", str(ticks)] + extra, capture_output=True, text=True, timeout=120)
            if res.returncode != 0:
                return {"note": f"driver failed ({res.returncode}): {res.stdout[-200:]}"}
            out[form] = json.loads(res.stdout.strip().splitlines()[-1])
        return {"p50": out["fused"]["p50_us"], "p99": out["fused"]["p99_us"], "mean": out["fused"]["mean_us"],
                "entry": "VariableSamplingMPCT::update + solveMPC -> vsmpc_tick (kinematics -> record -> solve in one "
                         "submission through the mapped staging buffer, one synchronisation)",
                "two_call_form": {"p50": out["two_call"]["p50_us"], "p99": out["two_call"]["p99_us"],
                                  "entry": "update -> vsmpc_kinematics_batch (synchronous), solveMPC -> vsmpc_solve_batch"},
                "what": f"batch=1, {ticks} ticks of a 45-state provider scenario, {out['fused']['solved']} solved; host wall-clock "
                        "of the two calls in a C++ process; reference: 2.18 ms per tick (poster, hardware unstated)"}
    except Exception as exc:  # pragma: no cover - needs g++ and the test fixtures
        return {"note": f"unavailable ({type(exc).__name__}: {exc})"}


def parity_sample(cfg_name: str, inputs: np.ndarray, x: np.ndarray, k: int = 8) -> float:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vsmpc_ref as ref
    rcfg = ref.paper_config() if cfg_name == "paper" else ref.horizon2x_config()
    worst = 0.0
    for b in np.linspace(0, len(inputs) - 1, k).astype(int):
        xr, _, _, _ = ref.solve_instance(rcfg, inputs[b])
        worst = max(worst, float(np.abs(x[b] - xr).max() / max(1.0, np.abs(xr).max())))
    return worst


# BASELINE.json configs[2], [3] (the per-GPU slice of the 32768-instance Monte-Carlo batch: 4096 per GPU, seeds as
# sharding.rank_inputs builds them for an 8-GPU node) and [4], at full size
EXTRA_CONFIGS = (
    {"name": "configs[2] batch=4096 take-off + disturbance seeds", "config": "paper", "workload": "takeoff",
     "batch": 4096, "steps": 40, "warmup": 4},
    {"name": "configs[3] Monte-Carlo (4 sigma), 4096 per GPU (rank slice of 32768 over 8 GPUs)", "config": "paper",
     "workload": "montecarlo", "batch": 4096, "steps": 40, "warmup": 4, "global_total": 32768},
    {"name": "configs[4] 2x horizon, batch=4096", "config": "horizon2x", "workload": "hover", "batch": 4096,
     "steps": 12, "warmup": 2},
)


def measure_extra(spec, pkg, synth, solver, sharding, dev, local_rank, rank, world, distributed, red_dev=None):
    """One extra configuration: every rank solves its own `batch`-instance slice; max-over-ranks timing."""
    import torch
    import torch.distributed as dist
    cfg = pkg.paper_config() if spec["config"] == "paper" else pkg.horizon2x_config()
    B = spec["batch"]
    if "global_total" in spec:       # configs[3]: rank r of an 8-GPU node owns instances [r*4096, (r+1)*4096)
        slots = spec["global_total"] // B
        inputs = sharding.rank_inputs(cfg, synth, spec["global_total"], rank % slots, slots, workload=spec["workload"])
    else:
        first, count = sharding.shard_range(B * world, rank, world)
        inputs = synth.make_batch(cfg, count, workload=spec["workload"], first_index=first)
    mpc = solver.BatchedVSMPC(cfg, device=local_rank, max_batch=B)
    d_in = torch.from_numpy(inputs).to(dev)
    d_x = torch.empty((B, cfg.n_var), dtype=torch.float64, device=dev)
    d_fm = torch.empty((B, 24), dtype=torch.float64, device=dev)
    d_st = torch.empty(B, dtype=torch.int32, device=dev)
    d_it = torch.empty(B, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    for _ in range(1 + spec["warmup"]):
        mpc.solve_device(d_in, d_x, d_fm, d_st, d_it, stream)
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    mpc.timing_begin(stream)
    for _ in range(spec["steps"]):
        mpc.solve_device(d_in, d_x, d_fm, d_st, d_it, stream)
    kernel_ms = mpc.timing_end(stream, spec["steps"])
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    red_dev = dev if red_dev is None else red_dev
    red = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=red_dev)
    lo = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)   # this rank's own wall-clock for the K steps
    cnt = torch.tensor([int((d_st == 1).sum().item()), int(d_it.sum().item()), 1], dtype=torch.int64, device=red_dev)
    if distributed:
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    rec = None
    if rank == 0:
        elapsed, kernel_ms = float(red[0].item()), float(red[1].item())
        tf = F_ALG[spec["config"]] * B / (kernel_ms * 1e-3) / 1e12
        x_host = d_x.cpu().numpy()
        rec = {"name": spec["name"], "config": spec["config"], "workload": spec["workload"], "batch_per_gpu": B,
               "n_gpus": world, "steps": spec["steps"], "value": B * world * spec["steps"] / elapsed, "unit": "solves/s",
               "ms_per_step": 1e3 * elapsed / spec["steps"], "kernel": mpc.kernel_name,
               "kernel_form": kernel_form(mpc, B, dev),
               "kernel_us_per_launch": kernel_ms * 1e3, "kernel_us_per_256": kernel_ms * 1e3 * 256 / B,
               "roofline_frac": tf / FP64_PEAK_TFLOPS, "achieved_tflops": tf,
               "solved": int(cnt[0].item()), "instances_per_step": B * world,
               "mean_active_set_iterations": float(cnt[1].item()) / (B * world),
               # ranks that took part in the all-reduce, and the spread of the per-rank rates (slowest .. fastest rank)
               "rccl_ranks": int(cnt[2].item()) if distributed else 1,
               "per_rank_solves_per_s": {"min": B * spec["steps"] / elapsed, "max": B * spec["steps"] / float(lo[0].item())},
               "parity_max_rel_err_vs_oracle": parity_sample(spec["config"], inputs, x_host, k=6)}
        ctr = counters_of(spec["config"], spec["workload"], B)
        rec.update(ctr)
        if ctr["executed_flops_per_solve"]:
            rec["executed_frac"] = ctr["executed_flops_per_solve"] * B / (kernel_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS
    mpc.close()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="instances per GPU (configs[1] = 256)")
    ap.add_argument("--workload", default="hover", choices=["hover", "takeoff", "montecarlo"])
    ap.add_argument("--config", default="paper", choices=["paper", "horizon2x"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs array (BASELINE configs[2..4])")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MPC path has no CPU fallback")
    # VSMPC_BENCH_BACKEND=gloo rehearses the N > 1 code path (slicing, barriers, reductions, the JSON line) on a box with
    # fewer GPUs than ranks: ranks share the devices and the reductions run on host tensors.  Not a measurement.
    backend = os.environ.get("VSMPC_BENCH_BACKEND", "nccl")
    rehearsal = backend != "nccl"
    if rehearsal:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = torch.device("cpu") if rehearsal else dev
    # under torch.distributed.run (RANK set) the RCCL group is created even for one rank, so that the same code path
    # (init, barrier, all_reduce) runs at every N
    distributed = world > 1 or "RANK" in os.environ
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if rehearsal:
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    solver = importlib.import_module(PKG + ".solver")
    sharding = importlib.import_module(PKG + ".sharding")
    import __graft_entry__ as ge
    if rank == 0:
        ge.build()                 # no-op when the in-tree libraries are current; only one rank may (re)build
    if distributed:
        dist.barrier()

    cfg = pkg.paper_config() if args.config == "paper" else pkg.horizon2x_config()
    B = args.batch
    default_line = args.config == "paper" and args.workload == "hover" and B == 256
    # N > 1 with the default flags: BASELINE.json configs[3] goes on the line -- 4096 Monte-Carlo instances (4x sigma) per
    # rank, rank r owning the global instances [r * 4096, (r + 1) * 4096) with its own seeds (sharding.rank_inputs)
    multi = world > 1 and default_line
    workload = args.workload
    if multi:
        B, workload = 4096, "montecarlo"
        inputs = sharding.rank_inputs(cfg, synth, B * world, rank, world, workload=workload)
        count = B
        if args.steps == 200:
            args.steps, args.warmup = 60, 6           # 60 x 0.43 ms: the same wall-clock order as 200 x 46 us
    else:
        first, count = sharding.shard_range(B * world, rank, world)   # contiguous slice, no exchange
        inputs = synth.make_batch(cfg, count, workload=workload, first_index=first)
    mpc = solver.BatchedVSMPC(cfg, device=local_rank, max_batch=count)

    d_in = torch.from_numpy(inputs).to(dev)
    d_x = torch.empty((count, cfg.n_var), dtype=torch.float64, device=dev)
    d_fm = torch.empty((count, 24), dtype=torch.float64, device=dev)
    d_st = torch.empty(count, dtype=torch.int32, device=dev)
    d_it = torch.empty(count, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def barrier():
        if distributed:
            dist.barrier()

    mpc.solve_device(d_in, d_x, d_fm, d_st, d_it, stream)   # set-up, not a step: loads the code object onto the device
    torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        mpc.solve_device(d_in, d_x, d_fm, d_st, d_it, stream)
    torch.cuda.synchronize(dev)

    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    mpc.timing_begin(stream)                     # HIP events on the launch stream bracket the same K launches
    for _ in range(args.steps):
        mpc.solve_device(d_in, d_x, d_fm, d_st, d_it, stream)
    kernel_ms = mpc.timing_end(stream, args.steps)
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    tmin = t.clone()
    solved = torch.tensor([int((d_st == 1).sum().item()), 1], dtype=torch.int64, device=red_dev)   # [solved, 1 per rank]
    kms = torch.tensor([kernel_ms], dtype=torch.float64, device=red_dev)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(solved, op=dist.ReduceOp.SUM)
        dist.all_reduce(kms, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    fastest = float(tmin.item())
    total_solved = int(solved[0].item())
    rccl_ranks = int(solved[1].item()) if distributed else 1      # every rank added 1: proves the collective saw N ranks
    kernel_ms = float(kms.item())

    if rank == 0:
        total = B * world
        value = total * args.steps / elapsed
        achieved_tflops = F_ALG[args.config] * count / (kernel_ms * 1e-3) / 1e12
        ctr = counters_of(args.config, workload, B)
        horizon = ("paper horizon/rates (nIter=17,nIterSmall=7,controlHorizon=12)" if args.config == "paper"
                   else "2x horizon (nIter=34,nIterSmall=14,controlHorizon=24)")
        what = (f"BASELINE configs[3]: {total} Monte-Carlo initial states (4x sigma) sharded over {world} GPUs, {B} per GPU, "
                f"per-rank seeds, {horizon}" if multi else
                f"batch={B} {workload} initial states per GPU, {horizon}")
        out = {
            # BASELINE.json's metric; `value` is the solves/s part, the p50 latency part is `latency_single_solve_us`
            "metric": "MPC solves/sec (whole node) + p50 single-solve latency, iRonCub paper horizon"
                      if args.config == "paper" else "MPC solves/sec (whole node) + p50 single-solve latency, 2x horizon",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": what,
                       "instances_total": total, "qp": f"{cfg.n_var} vars / {cfg.n_con} rows",
                       "parallelism": f"batch split over {world} GPU(s), no data-path collective"},
            "rccl_ranks": rccl_ranks, "collective_backend": ("rccl" if not rehearsal else backend + " (rehearsal, not a measurement)") if distributed else None,
            "per_rank_solves_per_s": {"min": count * args.steps / elapsed, "max": count * args.steps / fastest},
            # `frac` prices the METHOD-INDEPENDENT algorithmic FLOPs against the FP64 peak: an equivalent-throughput
            # figure.  `bound` says what the counters show the kernel is limited by -- dependent-instruction issue in lone
            # wavefronts, not the matrix pipe -- and `executed_*` what the wavefronts actually executed.
            "roofline": {"bound": ctr["bound"] or "issue", "peak_of": "fp64 mfma/fma", "achieved": achieved_tflops,
                         "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tflops / FP64_PEAK_TFLOPS, "frac_is": "algorithmic FLOPs (F_alg) / time / peak",
                         "executed_flops_per_solve": ctr["executed_flops_per_solve"],
                         "executed_frac": (ctr["executed_flops_per_solve"] * count / (kernel_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS
                                           if ctr["executed_flops_per_solve"] else None),
                         "wave_time_shares": ctr["wave_time_shares"],
                         "traffic": ctr["traffic"],
                         "mfma_busy_frac": ctr["mfma_busy_frac"], "counters_from": ctr["counters_from"],
                         "kernel": mpc.kernel_name, "kernel_form": kernel_form(mpc, B, dev),
                         "kernel_us_per_launch": kernel_ms * 1e3,
                         "alg_flops_per_solve": F_ALG[args.config], "alg_bytes_per_solve": BYTES_ALG[args.config],
                         "hbm_frac_informational": BYTES_ALG[args.config] * count / (kernel_ms * 1e-3) / 8.0e12},
            "solved": total_solved, "instances_per_step": total,
        }
        x_host = d_x.cpu().numpy()
        out["parity_max_rel_err_vs_oracle"] = parity_sample(args.config, inputs, x_host)
        if not args.no_latency:
            one = solver.BatchedVSMPC(cfg, device=local_rank, max_batch=1)
            one_host = solver.BatchedVSMPC(cfg, device=local_rank, max_batch=1)
            lat = []
            for i in range(1020):
                torch.cuda.synchronize(dev)
                a = time.perf_counter()
                one.solve_device(d_in[:1], d_x[:1], d_fm[:1], d_st[:1], d_it[:1], stream)
                torch.cuda.synchronize(dev)
                lat.append(time.perf_counter() - a)
            lat = np.array(lat[20:])
            out["latency_single_solve_us"] = {"p50": float(np.percentile(lat, 50) * 1e6),
                                              "p99": float(np.percentile(lat, 99) * 1e6),
                                              "entry": "vsmpc_solve_batch_device (record and outputs resident in HBM)",
                                              "what": "batch=1, 1000 repeats, host wall-clock incl. launch + sync"}
            one.close()
            rec1 = np.ascontiguousarray(inputs[:1])
            lat = []
            for i in range(1020):
                a = time.perf_counter()
                mpc_host_x = one_host.solve(rec1)
                lat.append(time.perf_counter() - a)
            lat = np.array(lat[20:])
            out["latency_host_entry_us"] = {"p50": float(np.percentile(lat, 50) * 1e6), "p99": float(np.percentile(lat, 99) * 1e6),
                                            "entry": "vsmpc_solve_batch (host record in, host results out, mapped staging buffer)",
                                            "what": "batch=1, 1000 repeats through the ctypes wrapper (solver.BatchedVSMPC.solve)"}
            one_host.close()
            if world == 1:
                out["latency_reference_surface_us"] = surface_latency()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config, inputs)
            out["cpu_structured"] = cpu_structured(args.config, inputs)

    # BASELINE.json configs[2..4] at full size, appended to the same line (every rank runs its own slice; `value` stays
    # on configs[1]).  Skipped when the caller picked a non-default workload itself.
    mpc.close()
    if default_line and not args.no_extra:
        extra = []
        specs = EXTRA_CONFIGS
        if multi:     # configs[3] is the headline here; the 256-per-GPU hover figure of the N = 1 line moves down here
            specs = ({"name": "configs[1] slice: batch=256 hover initial states per GPU", "config": "paper", "workload": "hover",
                      "batch": 256, "steps": 200, "warmup": 20},) + tuple(sp for sp in EXTRA_CONFIGS if "global_total" not in sp)
        for spec in specs:
            rec = measure_extra(spec, pkg, synth, solver, sharding, dev, local_rank, rank, world, distributed, red_dev)
            if rank == 0:
                extra.append(rec)
        if rank == 0:
            out["extra_configs"] = extra
    if rank == 0:
        print(json.dumps(out), flush=True)

    barrier()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
