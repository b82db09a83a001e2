#!/usr/bin/env python3
"""bench.py — GP fits/sec (Gram + LL^T + solve) at N = 16384 fp64 on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one `agp_fit_create` on one synthetic 3-D dataset whose inputs are
already resident in HBM: Gram of the measurement-wrapped features + target
variance on the diagonal, in-place LL^T, information vector K^-1 y and log|K|
(include/albatross/src/models/gp.hpp:281-294,61-69 in the reference).
Workload = BASELINE.json config 3's problem (3-D SquaredExponential(1,1) +
IndependentNoise(0.1), N = 16384, fp64), the size the metric is quoted on.

Rank 0 prints ONE JSON line (contract in the task statement) including
  roofline      for the dominant kernel (fp64 MFMA trailing update), from HIP
                events recorded on the library's stream around every launch
  cpu_baseline  the oracle (albatross-faithful port: serial Gram + unblocked
                pivoted LDL^T, 1 thread) timed on a bounded sample
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_TRAIN = 16384
DIM = 3
MFMA_F64_PEAK_TFLOPS = 78.6  # MI355X datasheet FP64 matrix; cross-checked by agp_mfma_f64_peak


def make_dataset(n, seed):
    """SURVEY.md §8d config 3 generator: X ~ U[0,10]^3, y = sum sin x_k + 0.1 cos(10 x_0)."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(0., 10., size=(n, DIM))
    y = np.sin(x).sum(axis=1) + 0.1 * np.cos(10. * x[:, 0])
    return x, y


def cpu_baseline(seconds_budget=30.0):
    """Oracle ("port") timed on host cores: albatross-faithful default = serial
    Gram + single-threaded unblocked pivoted LDL^T.  Bounded sample, scaled to
    fits/sec at N = 16384 by the N^3 law of the factorisation."""
    import albatross_amd as ab
    from oracle import oracle_py as orc
    cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
    n = 1024
    best = None
    while True:
        x, y = make_dataset(n, 44)
        t0 = time.perf_counter()
        fit = orc.OracleFit(cov, x, y)
        _ = fit.information
        dt = time.perf_counter() - t0
        del fit
        best = (n, dt)
        # next size costs ~8x (more once the matrix leaves the caches: x12); stop when it would blow the budget
        if dt * 12.0 > seconds_budget or n >= 8192:
            break
        n *= 2
    # one more sample at 1.5x the size when that still fits: the sample should be 10-30 s of CPU work
    if best[1] * 3.375 * 1.5 <= seconds_budget and best[0] < 8192:
        n = best[0] * 3 // 2
        x, y = make_dataset(n, 44)
        t0 = time.perf_counter()
        fit = orc.OracleFit(cov, x, y)
        _ = fit.information
        best = (n, time.perf_counter() - t0)
        del fit
    n, dt = best
    scaled = dt * (N_TRAIN / n) ** 3
    out = {"value": 1.0 / scaled, "unit": "fits/sec", "cores": 1, "kind": "port",
           "sample": f"one oracle fit (serial Gram + unblocked pivoted LDLT) at N={n}: {dt:.2f} s; "
                     f"scaled by (16384/{n})^3 to N=16384"}
    # BASELINE.md section 2, B2 "albatross-faithful, pooled": the Gram over all host cores (callers.hpp:134-166), the
    # factor unchanged (Eigen's LDLT has no parallel path) - the same fit once more with the pooled Gram
    try:
        import os as _os
        cores = _os.cpu_count() or 1
        xg, yg = make_dataset(n, 44)
        t0 = time.perf_counter()
        fit = orc.OracleFit(cov, xg, yg, threads=cores)
        _ = fit.information
        t_pooled = time.perf_counter() - t0
        del fit
        out["pooled_gram"] = {"value": 1.0 / (t_pooled * (N_TRAIN / n) ** 3), "unit": "fits/sec", "cores": cores,
                              "sample": f"the same oracle fit at N={n} with the Gram pooled over {cores} threads: {t_pooled:.2f} s"}
    except Exception as exc:  # noqa: BLE001 - context only
        out["pooled_gram"] = {"error": f"{type(exc).__name__}: {exc}"}
    # For context ("strong CPU"): the same fit on all host cores with a blocked,
    # multi-threaded LAPACK Cholesky (scipy) and a vectorised numpy Gram.  Not the reference's algorithm
    # (albatross factors with Eigen's unblocked single-threaded LDL^T), so it is reported beside `value`.
    try:
        import os as _os
        import scipy.linalg as sla
        m = 4096
        xs, ys = make_dataset(m, 44)
        t0 = time.perf_counter()
        sq = (xs * xs).sum(axis=1)
        d2 = np.maximum(sq[:, None] + sq[None, :] - 2.0 * (xs @ xs.T), 0.0)
        K = np.exp(-d2)
        K[np.diag_indices(m)] += 0.1 * 0.1
        t_gram = time.perf_counter() - t0
        t0 = time.perf_counter()
        c = sla.cho_factor(K, lower=True, overwrite_a=True, check_finite=False)
        sla.cho_solve(c, ys, check_finite=False)
        t_chol = time.perf_counter() - t0
        scaled_s = t_gram * (N_TRAIN / m) ** 2 + t_chol * (N_TRAIN / m) ** 3
        out["strong_cpu"] = {"value": 1.0 / scaled_s, "unit": "fits/sec", "cores": _os.cpu_count(),
                             "sample": f"numpy Gram ({t_gram:.2f} s, scaled by N^2) + LAPACK dpotrf/dpotrs via scipy on all "
                                       f"host cores ({t_chol:.2f} s, scaled by N^3) at N={m}"}
    except Exception as exc:  # noqa: BLE001 - context only
        out["strong_cpu"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=N_TRAIN)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-predict", action="store_true")
    ap.add_argument("--force-sharded", action="store_true", help="use the sharded-fit code path even on 1 GPU")
    ap.add_argument("--sharded-aux", action="store_true",
                    help="N > 1: also time ONE fit sharded over all ranks (RCCL panel broadcasts) after the timed "
                         "region; opt-in because a rank failing inside a collective would stall the other ranks")
    ap.add_argument("--multi-gpu", choices=["sharded", "replicas"], default="replicas",
                    help="N > 1: 'replicas' (default) = one independent fit per rank, no data-path collective, "
                         "value = aggregate fits/s (weak scaling); 'sharded' = value is ONE fit block-column-sharded "
                         "over all ranks with a panel broadcast per 512 columns over RCCL (strong scaling).  The mode "
                         "that is not `value` is still measured and reported in an auxiliary block.")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # BENCH_SINGLE_DEVICE=1 (testing the N > 1 code path on a one-GPU box): every rank uses GPU 0 and the
    # control collectives (barrier, max of the elapsed times) go over gloo - RCCL refuses two ranks per device.
    single_device = os.environ.get("BENCH_SINGLE_DEVICE") == "1"
    if single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    torch.cuda.init()  # torch's HIP runtime first, then the library's (albatross_amd/distributed.py)
    backend = "none"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if single_device else "nccl"
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    ctl_device = f"cuda:{local_rank}" if backend == "nccl" else "cpu"  # where the timing scalars are reduced

    import albatross_amd as ab
    from albatross_amd import _capi as capi

    n = args.n
    ctx = ab.Context(local_rank)
    lib = ctx._lib
    cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
    kh = ctx.kernel(cov)

    # inputs resident in HBM before the timed region
    x_h, y_h = make_dataset(n, 44 + rank)
    x_d = torch.from_numpy(x_h).to(f"cuda:{local_rank}")
    y_d = torch.from_numpy(y_h).to(f"cuda:{local_rank}")
    torch.cuda.synchronize()
    feats = capi.Features()
    feats.n, feats.dim, feats.n_scale_columns = n, DIM, 0
    feats.coords = x_d.data_ptr()
    feats.eq_id = None
    feats.scales = None
    feats.is_measurement = 0
    feats.location = capi.DEVICE

    sharded = (world > 1 and args.multi_gpu == "sharded") or args.force_sharded
    if sharded:
        from albatross_amd.distributed import HipBlockOps, ShardedGaussianProcessFit
        x_h, y_h = make_dataset(n, 44)  # every rank holds the same dataset: ONE fit over all ranks
        sfit = ShardedGaussianProcessFit(HipBlockOps(ctx, f"cuda:{local_rank}"), cov, block=512)

    def step():
        if sharded:
            sfit.fit(x_h, y_h)
            return
        h = C.c_void_p()
        st = lib.agp_fit_create(ctx._h, kh, C.byref(feats), C.c_void_p(y_d.data_ptr()), None, C.byref(h), None, None)
        if st != capi.AGP_OK:
            raise RuntimeError(f"agp_fit_create failed: {lib.agp_status_string(st).decode()} "
                               f"{lib.agp_last_error(ctx._h).decode()}")
        lib.agp_fit_destroy(h)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ctx.set_profiling(True)
    for _ in range(args.warmup):
        step()
    gemm_ms = gemm_flop = gemm_launches = 0.0
    gram_ms = factor_ms = solve_ms = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()  # agp_fit_create returns after its stream has drained
        gram_ms += ctx.stage_ms(0)
        factor_ms += ctx.stage_ms(1)
        solve_ms += ctx.stage_ms(2)
        gemm_ms += ctx.stage_ms(3)
        gemm_launches += ctx.stage_ms(4)
        gemm_flop += ctx.stage_ms(5)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=ctl_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- auxiliary (N > 1): the multi-GPU mode that is not `value` ----
    replicas = None
    sharded_aux = None
    if world > 1 and sharded:
        def replica_step():
            h = C.c_void_p()
            st = lib.agp_fit_create(ctx._h, kh, C.byref(feats), C.c_void_p(y_d.data_ptr()), None, C.byref(h), None, None)
            if st != capi.AGP_OK:
                raise RuntimeError("agp_fit_create failed")
            lib.agp_fit_destroy(h)
        replica_step()
        barrier()
        tr = time.perf_counter()
        for _ in range(3):
            replica_step()
        barrier()
        tr = time.perf_counter() - tr
        t = torch.tensor([tr], device=ctl_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        replicas = {"fits_per_sec": 3 * world / float(t.item()), "note": "one independent fit per GPU, no collective"}
    if world > 1 and not sharded and args.sharded_aux:
        # ONE fit sharded over all ranks (albatross_amd/distributed.py): block-column-cyclic LL^T with a
        # panel broadcast per 512 columns over RCCL.  A failure here is reported, it does not void `value`.
        try:
            from albatross_amd.distributed import HipBlockOps, ShardedGaussianProcessFit
            xs_h, ys_h = make_dataset(n, 44)
            sf = ShardedGaussianProcessFit(HipBlockOps(ctx, f"cuda:{local_rank}"), cov, block=512)
            sf.fit(xs_h, ys_h)
            barrier()
            tr = time.perf_counter()
            for _ in range(3):
                res = sf.fit(xs_h, ys_h)
            barrier()
            tr = time.perf_counter() - tr
            t = torch.tensor([tr], device=ctl_device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sharded_aux = {"single_fit_ms": 1e3 * float(t.item()) / 3, "fits_per_sec": 3 / float(t.item()),
                           "scaling": "strong",
                           "note": f"one N={n} fit block-column-sharded over {world} GPUs, RCCL panel broadcasts; "
                                   "one block column of look-ahead (DESIGN.md section 6)"}
        except Exception as exc:  # noqa: BLE001 - reported in the JSON line
            sharded_aux = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- secondary: predict points/sec at M = 4096 against one resident fit ----
    predict = None
    if rank == 0 and not args.no_predict and not sharded:
        m = 4096
        xs_h, _ = make_dataset(m, 43)
        xs_d = torch.from_numpy(xs_h).to(f"cuda:{local_rank}")
        out_d = torch.empty(2 * m, dtype=torch.float64, device=f"cuda:{local_rank}")
        fx = capi.Features()
        fx.n, fx.dim, fx.n_scale_columns = m, DIM, 0
        fx.coords = xs_d.data_ptr()
        fx.eq_id = None
        fx.scales = None
        fx.is_measurement = 0
        fx.location = capi.DEVICE
        h = C.c_void_p()
        st = lib.agp_fit_create(ctx._h, kh, C.byref(feats), C.c_void_p(y_d.data_ptr()), None, C.byref(h), None, None)
        assert st == capi.AGP_OK
        mean_p, var_p = C.c_void_p(out_d.data_ptr()), C.c_void_p(out_d.data_ptr() + 8 * m)

        def timed(fn, reps):
            fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / reps

        t_mean = timed(lambda: lib.agp_predict_mean(ctx._h, kh, h, C.byref(fx), mean_p, capi.DEVICE), 5)
        t_marg = timed(lambda: lib.agp_predict_marginal(ctx._h, kh, h, C.byref(fx), mean_p, var_p, capi.DEVICE), 3)
        lib.agp_fit_destroy(h)
        predict = {"m": m, "mean_pts_per_sec": m / t_mean, "marginal_pts_per_sec": m / t_marg,
                   "mean_ms": 1e3 * t_mean, "marginal_ms": 1e3 * t_marg}

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so
    # the per-launch figure comes from the committed rocprofv3 --pmc passes (profiles/r01/pmc_traffic.json,
    # FETCH_SIZE and WRITE_SIZE in separate runs, gfx950 x2 read correction applied); null if absent.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")) as fh:
            traffic = json.load(fh)["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        traffic = None

    if rank == 0:
        # sharded: one fit per step over all ranks; replicas: every rank fits its own dataset
        fits = args.steps if (sharded or world == 1) else args.steps * world
        achieved = (gemm_flop / 1e12) / (gemm_ms * 1e-3) if gemm_ms > 0 else 0.0
        out = {
            "metric": "GP fits/sec (Gram+Chol+solve) at N=16384 fp64",
            "value": fits / elapsed,
            "unit": "fits/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"dense GP fit, N={n}, 3-D SquaredExponential(1,1)+IndependentNoise(0.1), "
                                   "inputs resident in HBM (BASELINE config 3 problem)",
                       "parallelism": ("1 GPU" if world == 1 else
                                       (f"one fit block-column-sharded over {world} GPUs, panel broadcast per 512 columns (RCCL)"
                                        if sharded else f"{world} independent fits, one per GPU, no data-path collective"))},
            "roofline": {
                "bound": "mfma", "kernel": "agp::trailing_update_kernel (fp64 MFMA bulk trailing update C -= P P^T, K=512)",
                "achieved": achieved, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_F64_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_unit": "bytes per launch (rocprofv3 PMC passes, profiles/r01/pmc_traffic.json)",
                "launches_per_fit": gemm_launches / args.steps,
                "avg_launch_ms": gemm_ms / max(gemm_launches, 1.0),
                "flop_per_fit": gemm_flop / args.steps,
            },
            "stages_ms_per_fit": {"gram": gram_ms / args.steps, "factor": factor_ms / args.steps,
                                  "backward_solve": solve_ms / args.steps,
                                  "trailing_update_kernels": gemm_ms / args.steps},
        }
        if predict is not None:
            out["predict"] = predict
        if replicas is not None:
            out["replicas"] = replicas
        if sharded_aux is not None:
            out["sharded_single_fit"] = sharded_aux
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
