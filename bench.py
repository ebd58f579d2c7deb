"""Benchmark of the GP-transportation hot path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): posterior (mean + var + Jacobian) predictions/sec, N=8192 source points,
3-D, fp64.  A step = one fused predict pass (mean (M,3), variance (M,), Jacobian (M,3,3)) over one
batch of M = 500 000 synthetic queries already resident in HBM (configs[2]); with N GPUs every rank
runs the same per-GPU batch on its own query shard (configs[3]: 8 x 500k = 4M, weak scaling) after ONE
RCCL broadcast of the model fitted on rank 0.  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6      # MI355X datasheet fp64 matrix peak (not in MI355X_MICROARCH.md)


def flops_per_query(N, D, O, jvar):
    """Algorithmic work model of SURVEY §8d / BASELINE.md (exp = 1 flop)."""
    lin = N * (3 * D + 1 + 2 * O + 2 * D * O + D + 2)
    return (1 + D) * N * N + lin if jvar else N * N + lin


def var_kernel_flops_per_query(N, D, jvar):
    """Algorithmic flops of the dominant (variance) kernel alone: N^2 triangular multiply-add + 2N
    reduction + its k* row (3D N + N exp); x(1+D) columns with the Jacobian variance."""
    per_col = N * N + 2 * N
    return (1 + D) * per_col + N * (3 * D + 1) if jvar else per_col + N * (3 * D + 1)


def synthetic_sources(N, D=3, seed=0):
    """SURVEY §8d synthetic inputs: X ~ U[0,1]^D, Y = 0.05 sin(4X) + 0.01 N(0,1)."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (N, D))
    Y = 0.05 * np.sin(4 * X) + 0.01 * rng.standard_normal((N, D))
    return X, Y


def cpu_baseline(N, D, O, sample, jvar):
    """Times the CPU oracle (numpy/scipy restatement of the reference's algorithm, all host cores
    through BLAS) on a bounded sample of the same workload.  kind = "port"."""
    from oracle import gp_oracle as orc
    X, Y, Xq = orc.synthetic_problem(N, sample)
    c, ls, noise, jit = 0.1, np.array([0.1] * D), 1e-4, 1e-10
    t0 = time.perf_counter()
    L, a = orc.gpr_fit(X, Y, c, ls, noise, jit)
    t_fit = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.posterior_all_fast(Xq, X, L, a, c, ls, noise, want_jvar=jvar)
    t_pred = time.perf_counter() - t0
    out = {"value": sample / t_pred, "unit": "predictions/s", "cores": os.cpu_count(), "kind": "port",
           "sample": f"oracle.posterior_all_fast on {sample} of the queries at N={N} (mean+var+J"
                     f"{'+Jvar' if jvar else ''}), numpy/scipy BLAS threads; CPU fit (Gram+potrf+potrs) took {t_fit:.1f} s",
           "fit_s": t_fit}
    # The library the reference itself calls for fit / mean / std (gaussian_process.py:19-21, 37, 48): scikit-learn's
    # GaussianProcessRegressor with the reference's kernel and optimizer=None, same inputs, mean + std only (the
    # reference's Jacobian code is its own and does not exist on this box).  Reported beside the port, never as `value`.
    try:
        from sklearn.gaussian_process import GaussianProcessRegressor
        from sklearn.gaussian_process.kernels import RBF, ConstantKernel, WhiteKernel
        gpr = GaussianProcessRegressor(kernel=ConstantKernel(c) * RBF(ls) + WhiteKernel(noise), alpha=jit, optimizer=None)
        t0 = time.perf_counter()
        gpr.fit(X, Y)
        t_sk_fit = time.perf_counter() - t0
        t0 = time.perf_counter()
        gpr.predict(Xq, return_std=True)
        t_sk = time.perf_counter() - t0
        out["sklearn"] = {"fit_s": t_sk_fit, "mean_std_predictions_per_s": sample / t_sk,
                          "what": f"GaussianProcessRegressor(optimizer=None).fit + predict(return_std=True) on {sample} queries"}
    except ImportError:
        out["sklearn"] = None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-source", type=int, default=8192)
    ap.add_argument("--queries", type=int, default=500_000, help="queries per GPU per step")
    ap.add_argument("--jvar", action="store_true", help="also compute the Jacobian variance (mode J+Jvar)")
    ap.add_argument("--cpu-sample", type=int, default=2000, help="queries of the CPU baseline (0 = skip)")
    args = ap.parse_args()

    import torch
    from gaussian_process_transportation_amd import _lib
    from gaussian_process_transportation_amd.distributed import broadcast_model

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    _lib.require_gpu()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # GPT_BENCH_FORCE_DIST=1 runs the multi-rank code path (RCCL init, model broadcast, barriers, max-reduce) with
    # however many ranks there are, also one: the rehearsal available on a one-GPU box
    use_dist = world > 1 or os.environ.get("GPT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    N, D, O, M = args.n_source, 3, 3, args.queries
    h = _lib.Handle(local_rank)
    h.set_stream(torch.cuda.current_stream().cuda_stream)

    # ---- fit on rank 0, broadcast the factor (outside the timed region; reported)
    fit_ms = bcast_ms = None
    fit_timings = None
    if rank == 0:
        X, Y = synthetic_sources(N, D)
        ls = np.array([0.1] * D)
        h.fit(X, Y, ls, 0.1, 1e-4, 1e-10)                  # first call: allocations + code load
        t0 = time.perf_counter()
        h.fit(X, Y, ls, 0.1, 1e-4, 1e-10)
        fit_ms = (time.perf_counter() - t0) * 1e3
        fit_timings = h.fit_timings()
    bcast_bytes = 0
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bcast_bytes = broadcast_model(h, fitted=(rank == 0), src=0, device=dev)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - t0) * 1e3

    # ---- this rank's query shard, resident in HBM
    xq_host = np.random.default_rng(1 + rank).uniform(-0.1, 1.1, (M, D))
    xq = torch.from_numpy(xq_host).to(dev)
    mean = torch.empty((M, O), dtype=torch.float64, device=dev)
    var = torch.empty((M,), dtype=torch.float64, device=dev)
    J = torch.empty((M, O, D), dtype=torch.float64, device=dev)
    Jvar = torch.empty((M, D), dtype=torch.float64, device=dev) if args.jvar else None

    h.reserve(M, args.jvar)          # library scratch of the timed calls: allocated here, not inside the first step

    def step():
        h.predict_all_dev(xq.data_ptr(), M, mean.data_ptr(), var.data_ptr(), J.data_ptr(),
                          Jvar.data_ptr() if Jvar is not None else 0, 0)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    h.set_profiling(True)
    var_ms, mj_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        t = h.predict_timings()          # waits for this step's events only (the next launch follows at once)
        var_ms.append(t["var_ms"]); mj_ms.append(t["mean_jac_ms"])
    sync_all()
    elapsed = time.perf_counter() - t0
    h.set_profiling(False)
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- sanity of the timed outputs (finite, variance within [0, c+noise])
    ok = bool(torch.isfinite(mean).all() and torch.isfinite(J).all() and torch.isfinite(var).all()
              and float(var.min()) >= 0.0 and float(var.max()) <= 0.1 + 1e-4 + 1e-12)
    if not ok and not os.environ.get("GPT_BENCH_ABLATE"):     # timing-only ablation builds compute wrong values
        raise SystemExit("bench: non-finite or out-of-range outputs")

    if rank == 0:
        total_q = world * M * args.steps
        value = total_q / elapsed
        kern_ms = float(np.mean(var_ms))
        kflops = var_kernel_flops_per_query(N, D, args.jvar) * M
        achieved = kflops / (kern_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("jvar" if args.jvar else "j", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "posterior (mean+var+Jacobian) preds/sec, N=8192 source pts, 3-D fp64",
            "value": value, "unit": "predictions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"N={N} source pts, M={M} queries per GPU per step, D=O=3, "
                                   f"mean+var+Jacobian{'+Jacobian variance' if args.jvar else ''}, RBF GP, fp64"
                                   + (f"; {world}x{M} queries sharded, one RCCL broadcast of the factor" if world > 1 else ""),
                       "n_source": N, "queries_per_gpu": M, "mode": "J+Jvar" if args.jvar else "J"},
            "flops_per_query": flops_per_query(N, D, O, args.jvar),
            "achieved_tflops_whole_path": value * flops_per_query(N, D, O, args.jvar) / 1e12 / world,
            "roofline": {"bound": "mfma", "kernel": "k_var (variance / Jacobian-variance triangular MFMA GEMM)",
                         "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic,
                         "kernel_ms": kern_ms, "mean_jac_kernel_ms": float(np.mean(mj_ms))},
            "fit_ms": fit_ms, "fit_phases_ms": fit_timings, "bcast_ms": bcast_ms, "bcast_bytes": bcast_bytes,
        }
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(N, D, O, args.cpu_sample, args.jvar)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    h.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
