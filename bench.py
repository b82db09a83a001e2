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

N > 1: `value` is ONE fit of the same problem row-block-sharded over the N GPUs (RCCL inside the library; strong
scaling); N independent fits ("replicas", weak scaling) are reported in an auxiliary block.

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


def mt19937_uniform(seed, count, lo=0., hi=10.):
    """`std::mt19937 gen(seed); std::uniform_real_distribution<double> dist(lo, hi)` of libstdc++, as the reference's
    benchmarks draw their features (benchmarks/bench_utils.h:25-34): generate_canonical<double, 53> takes two 32-bit
    draws, low word first.  Bit-identical to the compiled generator (tests/test_oracle_golden.py checks it on
    tests/golden/bench512.json)."""
    rs = np.random.RandomState(seed)  # init_genrand(seed) == std::mt19937(seed)
    raw = np.frombuffer(rs.bytes(8 * count), dtype="<u4").astype(np.float64)
    u = (raw[0::2] + raw[1::2] * 4294967296.0) / 18446744073709551616.0
    return lo + (hi - lo) * np.minimum(u, np.nextafter(1.0, 0.0))


def make_dataset(n, seed):
    """SURVEY.md section 8d config 3 generator: X ~ U[0,10]^3 from mt19937(seed), row-major;
    y = sum_k sin x_k + 0.1 cos(10 x_0)."""
    x = mt19937_uniform(seed, n * DIM).reshape(n, DIM)
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
        cores = os.cpu_count() or 1
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
    # For context ("strong CPU"): the same fit with a blocked, multi-threaded LAPACK Cholesky (scipy) and a
    # vectorised numpy Gram.  Not the reference's algorithm (albatross factors with Eigen's unblocked single-threaded
    # LDL^T), so it is reported beside `value`.  Timed at N >= 8192 with the BLAS pool pinned to the physical cores
    # (a small matrix on every hardware thread measures oversubscription, not the factorisation).
    try:
        import scipy.linalg as sla
        from threadpoolctl import threadpool_limits
        try:
            usable = len(os.sched_getaffinity(0))
        except AttributeError:
            usable = os.cpu_count() or 2
        candidates = sorted({t for t in (8, 16, 32, max(1, min(usable // 2, 64))) if t <= usable} or {1})

        def strong_fit(m):
            xs, ys = make_dataset(m, 44)
            t0 = time.perf_counter()
            sq = (xs * xs).sum(axis=1)
            K = sq[:, None] + sq[None, :] - 2.0 * (xs @ xs.T)
            np.maximum(K, 0.0, out=K)
            np.negative(K, out=K)
            np.exp(K, out=K)
            K[np.diag_indices(m)] += 0.1 * 0.1
            t_gram = time.perf_counter() - t0
            t0 = time.perf_counter()
            c = sla.cho_factor(K, lower=True, overwrite_a=True, check_finite=False)
            sla.cho_solve(c, ys, check_finite=False)
            return t_gram, time.perf_counter() - t0

        # the pool size that factors fastest on THIS box (a container's CPU quota can be far below its visible cores)
        best_t, threads = None, candidates[0]
        for t in candidates:
            with threadpool_limits(limits=t):
                strong_fit(1024)
                tg, tc = strong_fit(4096)
            if best_t is None or tg + tc < best_t:
                best_t, threads = tg + tc, t
        with threadpool_limits(limits=threads):
            t_gram, t_chol = strong_fit(8192)
            if t_gram + t_chol < 3.0:
                t_gram, t_chol = strong_fit(N_TRAIN)
                out["strong_cpu"] = {"value": 1.0 / (t_gram + t_chol), "unit": "fits/sec", "cores": threads,
                                     "sample": f"numpy Gram ({t_gram:.2f} s) + LAPACK dpotrf/dpotrs via scipy ({t_chol:.2f} s) "
                                               f"at N={N_TRAIN} itself, BLAS pool pinned to {threads} threads (fastest of {candidates} at N=4096; {usable} usable hardware threads)"}
            else:
                scaled_s = t_gram * 4.0 + t_chol * 8.0
                out["strong_cpu"] = {"value": 1.0 / scaled_s, "unit": "fits/sec", "cores": threads,
                                     "sample": f"numpy Gram ({t_gram:.2f} s, x4) + LAPACK dpotrf/dpotrs via scipy ({t_chol:.2f} s, "
                                               f"x8) at N=8192, BLAS pool pinned to {threads} threads (fastest of {candidates} at N=4096; {usable} usable hardware threads)"}
    except Exception as exc:  # noqa: BLE001 - context only
        out["strong_cpu"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=N_TRAIN,
                    help="training points (default 16384 = BASELINE config 3; 32768 / 65536: sizes where sharding one fit pays)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-predict", action="store_true")
    ap.add_argument("--force-sharded", action="store_true", help="N = 1: time the sharded entry point (one rank, no transport)")
    ap.add_argument("--multi-gpu", choices=["sharded", "replicas"], default="sharded",
                    help="N > 1: 'sharded' (default) = `value` is ONE fit row-block-sharded over all ranks (RCCL broadcast + "
                         "all-gather per block column, strong scaling); 'replicas' = one independent fit per rank, no "
                         "data-path collective (weak scaling).  The other mode is measured too and reported in an auxiliary block.")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # one process per GPU: N > 1 is launched as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`;
        # nothing has touched the GPU yet, the launcher can simply be re-run
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks with "
                         f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
                         f"bench.py --gpus {args.gpus} ...`")

    import datetime
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # BENCH_SINGLE_DEVICE=1 (testing the N > 1 code path on a one-GPU box): every rank uses GPU 0 and the collectives of
    # the sharded fit go over gloo through the library's callback transport - RCCL refuses two ranks per device.
    single_device = os.environ.get("BENCH_SINGLE_DEVICE") == "1"
    if single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    torch.cuda.init()  # torch's HIP runtime first, then the library's
    if world > 1:
        # torch.distributed is the CONTROL plane only (rendezvous, exchange of the 128-byte RCCL id): gloo, finite timeout.
        # The data path - and the barrier / max-over-ranks of the timing - run on the library's own RCCL communicator.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=180))

    import albatross_amd as ab
    from albatross_amd import _capi as capi
    from albatross_amd.distributed import Communicator, ShardedGaussianProcessFit

    n = args.n
    ctx = ab.Context(local_rank)
    lib = ctx._lib
    cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
    kh = ctx.kernel(cov)
    comm = None
    transport = "none"
    if world > 1:
        transport = "callbacks" if single_device else "rccl"
        if transport == "rccl":
            # RCCL bootstrap can fail on a box whose network set-up it does not like; every rank must then take the same
            # exit.  The fallback is the SAME sharded fit with its collectives staged over gloo (slow, and said so in
            # the JSON line) - a measured line beats none.
            err = ""
            try:
                comm = Communicator.from_torch(ctx, transport="rccl")
            except Exception as exc:  # noqa: BLE001
                err = f"{type(exc).__name__}: {exc}"
            flag = torch.tensor([1.0 if comm is None else 0.0])
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if flag.item() > 0:
                if comm is not None:
                    comm.close()
                sys.stderr.write(f"bench.py rank {rank}: RCCL communicator not available ({err or 'failed on another rank'}); "
                                 "falling back to collectives staged over gloo\n")
                transport = "callbacks (RCCL communicator could not be created: collectives staged through host memory over gloo)"
                comm = Communicator.from_torch(ctx, transport="callbacks")
        else:
            comm = Communicator.from_torch(ctx, transport="callbacks")
        assert comm.world == world and comm.rank == rank

    sharded = (world > 1 and args.multi_gpu == "sharded") or (world == 1 and args.force_sharded)
    # inputs resident in HBM before the timed region.  Sharded: every rank holds the SAME dataset (one fit over all
    # ranks); replicas: one dataset per rank.
    x_h, y_h = make_dataset(n, 44 if sharded or world == 1 else 44 + rank)
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
    sfit = ShardedGaussianProcessFit(ctx, cov, comm) if (sharded or world > 1) else None

    def replica_step():
        h = C.c_void_p()
        st = lib.agp_fit_create(ctx._h, kh, C.byref(feats), C.c_void_p(y_d.data_ptr()), None, C.byref(h), None, None)
        if st != capi.AGP_OK:
            raise RuntimeError(f"agp_fit_create failed: {lib.agp_status_string(st).decode()} "
                               f"{lib.agp_last_error(ctx._h).decode()}")
        lib.agp_fit_destroy(h)

    def sharded_step():
        sfit.fit(None, None, features_struct=feats, device_targets=y_d.data_ptr())

    step = sharded_step if sharded else replica_step

    def barrier():
        torch.cuda.synchronize()
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        return float(comm.all_reduce([v], "max")[0]) if comm is not None else v

    try:
        ctx.set_profiling(True)
        for _ in range(args.warmup):
            step()
        gemm_ms = gemm_flop = gemm_launches = 0.0
        gram_ms = factor_ms = solve_ms = 0.0
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()  # returns after its streams have drained
            if sharded:
                gram_ms += sfit.stage(0)
                factor_ms += sfit.stage(1)
                gemm_ms += sfit.stage(3)
                gemm_launches += sfit.stage(4)
                gemm_flop += sfit.stage(5)
            else:
                gram_ms += ctx.stage_ms(0)
                factor_ms += ctx.stage_ms(1)
                solve_ms += ctx.stage_ms(2)
                gemm_ms += ctx.stage_ms(3)
                gemm_launches += ctx.stage_ms(4)
                gemm_flop += ctx.stage_ms(5)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)

        # ---- auxiliary (N > 1): the multi-GPU mode that is not `value` ----
        aux = None
        if world > 1:
            other = replica_step if sharded else sharded_step
            if not sharded:  # the sharded mode needs the same dataset on every rank
                x_h, y_h = make_dataset(n, 44)
                x_d.copy_(torch.from_numpy(x_h))
                y_d.copy_(torch.from_numpy(y_h))
                torch.cuda.synchronize()
            other()
            barrier()
            tr = time.perf_counter()
            for _ in range(3):
                other()
            barrier()
            tr = max_over_ranks(time.perf_counter() - tr)
            if sharded:
                aux = {"replicas": {"fits_per_sec": 3 * world / tr, "scaling": "weak",
                                    "note": "one independent fit per GPU (every rank its own copy of the problem), no collective"}}
            else:
                aux = {"sharded_single_fit": {"single_fit_ms": 1e3 * tr / 3, "fits_per_sec": 3 / tr, "scaling": "strong",
                                              "note": f"one N={n} fit row-block-sharded over {world} GPUs"}}
    except Exception as exc:  # noqa: BLE001
        # a failed or timed-out collective leaves the other ranks inside theirs: report, and leave with a non-zero code at
        # once (no destructors, no re-exec: the launcher starts fresh children)
        sys.stderr.write(f"bench.py rank {rank}: {type(exc).__name__}: {exc}\n")
        sys.stderr.flush()
        os._exit(3)

    # ---- secondary: predict points/sec at M = 4096 against one resident fit ----
    predict = None
    if rank == 0 and not args.no_predict and world == 1 and not sharded:
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

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the per-launch figure
    # comes from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs, gfx950 x2 read
    # correction applied) of the single-GPU run; null if absent or not applicable.
    traffic = traffic_src = None
    if world == 1 and not sharded and n == N_TRAIN:
        for rnd in ("r02", "r01"):
            try:
                with open(os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")) as fh:
                    traffic = json.load(fh)["traffic_bytes_per_launch"]
                traffic_src = f"profiles/{rnd}/pmc_traffic.json"
                break
            except (OSError, KeyError, ValueError):
                continue

    if rank == 0:
        # sharded: one fit per step over all ranks; replicas: every rank fits its own dataset
        fits = args.steps if (sharded or world == 1) else args.steps * world
        achieved = (gemm_flop / 1e12) / (gemm_ms * 1e-3) if gemm_ms > 0 else 0.0  # 0: no launch of that kernel at this size
        if sharded and world > 1:
            parallelism = (f"ONE fit row-block-sharded (512-row blocks, snake-cyclic) over {world} GPUs: per block column an RCCL "
                           "broadcast of the diagonal block and an all-gather of the panel, one block column of look-ahead")
            kernel_name = "agp::gemm_nt_sub_kernel (fp64 MFMA updates of rank 0's own row blocks, K=512)"
        else:
            parallelism = "1 GPU" if world == 1 else f"{world} independent fits, one per GPU, no data-path collective"
            kernel_name = "agp::trailing_update_kernel (fp64 MFMA bulk trailing update C -= P P^T, K=512)"
        out = {
            "metric": f"GP fits/sec (Gram+Chol+solve) at N={n} fp64",
            "value": fits / elapsed,
            "unit": "fits/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if (sharded and world > 1) else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"dense GP fit, N={n}, 3-D SquaredExponential(1,1)+IndependentNoise(0.1), features from "
                                   "mt19937(44), inputs resident in HBM (BASELINE config 3 problem)",
                       "parallelism": parallelism,
                       "transport": transport, "n_ranks": (comm.world if comm is not None else 1)},
            "roofline": {
                "bound": "mfma", "kernel": kernel_name,
                "achieved": achieved, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_F64_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_unit": f"bytes per launch (rocprofv3 PMC passes, {traffic_src})" if traffic_src else None,
                "launches_per_fit": gemm_launches / args.steps,
                "avg_launch_ms": gemm_ms / max(gemm_launches, 1.0),
                "flop_per_fit": gemm_flop / args.steps,
                # context, not the contract's `peak`: what back-to-back independent v_mfma_f64_16x16x4_f64 sustain on this
                # part (profiles/r01/microbench_fp64.txt: one instruction per ~100 cycles and SIMD at full clock; r02
                # mfma_rates.txt on random operands: 39-40) and what this kernel does alone at M = 15872 (48.4)
                "measured_pipe_ceiling_tflops": 48.0,
            },
            "stages_ms_per_fit": {"gram": gram_ms / args.steps, "factor": factor_ms / args.steps,
                                  "backward_solve": solve_ms / args.steps,
                                  "trailing_update_kernels": gemm_ms / args.steps},
        }
        if sharded:
            out["stages_ms_per_fit"]["note"] = "sharded entry point: `factor` is host wall time of factorisation + both substitutions"
        if predict is not None:
            out["predict"] = predict
        if aux is not None:
            out.update(aux)
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.barrier()
        comm.close()
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
