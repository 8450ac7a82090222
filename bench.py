#!/usr/bin/env python
"""bench.py — headline measurement of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, or — when
    WORLD_SIZE is not set — bench.py starts its N ranks itself the same way; it exits non-zero rather than fall back to
    the single-GPU headline when N GPUs are not there or WORLD_SIZE disagrees with --gpus)

Metric (BASELINE.json): soft-min pairs/s at N=M=1e6, D=3, fp32 (+ Sinkhorn wall-clock, + roofline).
A *pair* is one evaluation of exp(h_j - C(x_i,y_j)/eps) inside one soft-min reduction.

* N = 1 (the headline): one "step" = one ``glhip_softmin_fwd`` launch over one synthetic problem of 1e6 x 1e6 points
  in 3D (1e12 pairs), inputs resident in HBM.  Next to it, outside the timed region: the other reductions of the hot
  path timed with HIP events (``kernels``), the end-to-end losses of the BASELINE configs (``sinkhorn_wallclock``) and
  the CPU baseline (``cpu_baseline``).
* N > 1: BASELINE configs[3] — the batch of B = 256 problems of 4096 x 4096 bf16 points, sharded over the ranks by
  ``geomloss_amd.distributed.ShardedSamplesLoss`` (one process per GPU, explicit ``diameter``, no data-path
  collective); one step = one whole batched Sinkhorn loss (44 soft-min reductions per problem, fused four to a launch)
  + the RCCL all-reduce of the scalar loss.  Total work is fixed (``"scaling": "strong"``); value = soft-min pairs of
  the whole batch per second, max-over-ranks time.  The N = 1 line carries the same workload on one GPU as
  ``sharded_batch_reference`` so that the curve has its single-GPU point.
* roofline (N = 1): the kernel is bound by VALU issue, not by HBM — per pair it needs exactly one ``v_exp_f32`` (quarter
  rate: 8 cycles per wave64 instruction) and one ``v_add_f32`` (2 cycles) besides the MFMA that forms the exponent:
  peak = 256 CU x 4 SIMD x 2.4 GHz x 64 pairs / 10 cycles = 1.573e13 pairs/s.  ``achieved`` = pairs per launch /
  mean launch duration measured here with HIP events on the launch stream.  The dense-equivalent HBM figure that
  BASELINE.json asks for (4 algorithmic bytes per pair) is kept in ``hbm_dense_equivalent``; it
  exceeds 1 because the kernel never streams that matrix.  ``traffic`` = HBM bytes per launch of the dominant kernel from two
  rocprofv3 PMC passes that this process runs over a minimal copy of its own command (``measure_traffic``; null if the
  profiler is not there); the builder's fuller PMC summary of the same command stays under ``builder_pmc`` with its source.
* cpu_baseline (rank 0, N = 1): PyTorch-CPU port of the reference's tensorized Sinkhorn
  (oracle/tensorized_torch.py, pinned to the reference by tests/test_oracle_golden.py::test_torch_port_matches_reference)
  on BASELINE configs[0] exactly (N=M=2000, 2D, fp32) and at N=M=5000 3D.

Rank 0 prints ONE JSON line on stdout; everything else goes to stderr.
"""

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
CLOCK_HZ, SIMDS = 2.4e9, 256 * 4
# VALU issue model of one pair (guide: a wave64 VALU instruction issues over 2 cycles on a SIMD-32; transcendentals
# run at quarter rate): v_exp_f32 8 cycles + v_add_f32 2 cycles per 64 pairs.
NOMINAL_CYCLES_PER_64_PAIRS = 10.0
VALU_PEAK_PAIRS_PER_S = SIMDS * CLOCK_HZ * 64 / NOMINAL_CYCLES_PER_64_PAIRS
# measured on this part (tools/ubench, profiles/r01_ubench_pipes.txt): that exp2 + add stream alone runs at 12.5
# cycles per 64 pairs (v_exp_f32 8.2-9.7, v_add_f32 2.5-3.1), the kernel's bare inner loop (MFMA pair + stream) at 13.5
MEASURED_STREAM_CYCLES = 12.5
PMC_SUMMARY = os.path.join("profiles", "r06_pmc_softmin.json")
# The launch the headline times: what SamplesLoss("sinkhorn", blur=.05) runs on unit-cube clouds — exponents from two f16 pieces per
# coordinate, ONE v_mfma_f32_32x32x16_f16 per 1024 pairs (GLHIP_FLAG_F16X2; in range here: diameter^2 / eps = 1200, the flag's
# contract allows ~1.5e5).  The default layout of a raw C-ABI call (bf16 x 3, two MFMAs) is timed next to it (`bf16x3_layout`).
HEADLINE_FLAGS = 256
DOMINANT_KERNELS = ("xd_fwd_kernel", "softmin_fwd_x32_kernel")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


_JSON_OUT = None


def claim_stdout():
    """Rank 0 prints ONE JSON line on stdout — but libraries write there too (RCCL prints a five-line version banner on fd 1 when the
    process group goes away).  Keep a private copy of the real stdout for the line and point fd 1 at stderr for everybody else."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(obj):
    out = _JSON_OUT if _JSON_OUT is not None else sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


def make_problem(n, dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.rand(n, 3, generator=g).to(dev)
    y = torch.rand(n, 3, generator=g).to(dev)
    eps = 0.05**2
    # dual vector of a late Sinkhorn iteration: log-weights + potential / eps
    h = (torch.full((n,), -math.log(n)) + 0.01 * torch.randn(n, generator=g) / eps).to(dev)
    return x[None].contiguous(), y[None].contiguous(), h[None].contiguous(), eps


def event_ms(fn, reps, warmup=1):
    """Mean duration of ``fn()`` in ms, HIP events recorded on torch's current stream (the stream the C-ABI launches on)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    b = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    for k in range(reps):
        a[k].record()
        fn()
        b[k].record()
    torch.cuda.synchronize()
    return sum(s.elapsed_time(e) for s, e in zip(a, b)) / reps


def cpu_baseline(budget_s=10.0):
    from oracle.tensorized_torch import sinkhorn_tensorized_cpu

    cores = os.cpu_count() or 1
    # BASELINE configs[0] exactly: torch.manual_seed(0); x, y = rand(2000, 2) (tests/golden/make_golden.py, cfg1)
    torch.manual_seed(0)
    x1, y1 = torch.rand(2000, 2)[None], torch.rand(2000, 2)[None]
    g = torch.Generator().manual_seed(0)
    x5, y5 = torch.rand(1, 5000, 3, generator=g), torch.rand(1, 5000, 3, generator=g)
    # PyTorch's CPU ops do not scale to every core of a many-socket host: pick the fastest thread count from a sweep AT THE TIMED SIZE
    # (round-5 advice: a sweep at N = 1000 need not find the optimum of N = 2000).  One loss per thread count; the thread counts are
    # visited in increasing order and the sweep stops as soon as one takes 4x the best so far or more than `cap_s` seconds (over-subscribed,
    # 256 threads take 34 s for ONE loss at N = 2000 — 600x the 16-thread time); the all-cores figure is then measured on the first
    # 1000 points, which is what fits a bounded leg.
    best_t, best_threads, sweep, cap_s = None, 1, {}, 6.0
    for threads in sorted({min(cores, t) for t in (8, 16, 32, 64, cores)}):
        torch.set_num_threads(threads)
        sinkhorn_tensorized_cpu(x1[:, :300], y1[:, :300])   # warm the pool
        cnt = {}
        t0 = time.perf_counter()
        sinkhorn_tensorized_cpu(x1, y1, count=cnt)
        dt = time.perf_counter() - t0
        sweep[threads] = (cnt["softmin_calls"] * 2000.0 * 2000.0 / dt, dt)
        if best_t is None or dt < best_t:
            best_t, best_threads = dt, threads
        if dt > 4.0 * best_t or dt > cap_s:
            break
    all_n = 2000
    if cores not in sweep:      # every logical core: bounded sample
        all_n = 1000
        torch.set_num_threads(cores)
        xs, ys = x1[:, :all_n].contiguous(), y1[:, :all_n].contiguous()
        sinkhorn_tensorized_cpu(x1[:, :300], y1[:, :300])
        cnt = {}
        t0 = time.perf_counter()
        sinkhorn_tensorized_cpu(xs, ys, count=cnt)
        dt = time.perf_counter() - t0
        sweep[cores] = (cnt["softmin_calls"] * float(all_n) * all_n / dt, dt)
    torch.set_num_threads(best_threads)

    def timed(x, y, max_runs, budget):
        cnt, times, t_all = {}, [], time.perf_counter()
        while len(times) < max_runs and (not times or time.perf_counter() - t_all < budget):
            t0 = time.perf_counter()
            loss = sinkhorn_tensorized_cpu(x, y, count=cnt)
            times.append(time.perf_counter() - t0)
        t = sorted(times)[len(times) // 2]
        return cnt["softmin_calls"] * x.shape[1] * y.shape[1] / t, t, len(times), cnt["softmin_calls"], float(loss)

    v1, t1, n1, c1, l1 = timed(x1, y1, 9, budget_s * 0.5)
    v5, t5, n5, c5, _ = timed(x5, y5, 3, budget_s * 0.5)
    # north star: "core count stated" — every logical core as well: the sweep's entry for `cores` threads (same 36 soft-mins, N = M = 1000)
    all_cores = {"value": sweep[cores][0], "unit": "pairs/s", "cores": cores, "seconds": sweep[cores][1], "runs": 1,
                 "sample": f"N = M = {all_n} of BASELINE configs[0], one run"}
    thread_sweep = {str(k): v[0] for k, v in sorted(sweep.items())}
    return {
        "value": v1, "unit": "pairs/s", "cores": best_threads, "cores_total": cores, "all_cores": all_cores, "thread_sweep_pairs_per_s": thread_sweep,
        "kind": "port",
        "sample": f"BASELINE configs[0] exactly: PyTorch-CPU tensorized SamplesLoss('sinkhorn',p=2,blur=.05) forward, N=M=2000 2D fp32, "
                  f"seed 0 ({c1} dense soft-mins, median of {n1} runs, {t1:.3f} s each, loss {l1:.7e}; reference value 1.9925134e-04); "
                  f"{best_threads} torch threads = fastest of a sweep on a host with {cores} logical cores. "
                  "Port = oracle/tensorized_torch.py (the Python reference cannot travel to the GPU box)",
        "n5000_3d": {"value": v5, "unit": "pairs/s", "seconds": t5, "runs": n5, "softmin_calls": c5},
        "note": "tensorized cannot run at N=1e6 (4 TB per cost matrix): no CPU number exists for the headline size",
    }


def measure_traffic(points, timeout_s=120):
    """HBM traffic of the dominant kernel, measured HERE: two `rocprofv3 --pmc <counter> --kernel-trace` passes (FETCH_SIZE and
    WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, rocprofv3 PMC slots) over a minimal run of this very command
    (`bench.py --no-extras`: warm-up + 2 timed launches).  Counters are KiB per dispatch, averaged over the launches of
    the dominant kernel (xd_fwd_kernel<.., D = 3, f16 x 2>); FETCH_SIZE is doubled (gfx950 counts the 128-byte requests of a wide coalesced read at 64 bytes: the
    guide's correction — our column loads are such reads, the doubled figure is the upper estimate).  Returns a dict or None."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    out = tempfile.mkdtemp(prefix="glhip_pmc_", dir="/tmp")
    got = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-extras", "--points", str(points)]
            env = dict(os.environ, TMPDIR="/tmp")
            env.pop("WORLD_SIZE", None)
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
                con = sqlite3.connect(db)
                rows = con.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                                   "group by kernel_name, counter_name").fetchall()
                for k, cn, n, avg, dur in rows:
                    if any(name in k for name in DOMINANT_KERNELS) and cn == counter:
                        got[counter] = {"kib_per_launch": avg, "launches": n, "kernel_ns_under_profiler": dur}
        if "FETCH_SIZE" not in got or "WRITE_SIZE" not in got:
            return None
        f, w = got["FETCH_SIZE"]["kib_per_launch"], got["WRITE_SIZE"]["kib_per_launch"]
        return {"bytes_per_launch": (2.0 * f + w) * 1024.0, "bytes_per_launch_uncorrected": (f + w) * 1024.0, "counters": got,
                "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace, two passes over `bench.py --no-extras`, run by this "
                          "process; the reducing kernel only (the pack and merge kernels of the same call move ~0.2 GB more)"}
    except Exception as e:      # a profiler hiccup must not cost the bench line
        log(f"[bench] traffic leg failed: {e!r}")
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def hot_path_kernels(dev, n=1_000_000):
    """The other reductions of the hot path at N = M = n, each timed with HIP events around the C-ABI call."""
    from geomloss_amd import hip

    x, y, h, eps = make_problem(n, dev, seed=7)
    g = torch.randn(1, n, device=dev)
    v = torch.rand(1, n, device=dev) / n
    blur = 0.05
    out = hip.softmin_fwd_raw(x, y, h, eps, 2)
    pairs = float(n) * n
    res = {}

    def add(name, fn, reps=3):
        ms = event_ms(fn, reps)
        res[name] = {"ms": ms, "pairs_per_s": pairs / (ms * 1e-3)}
        log(f"[bench] {name}: {ms:.2f} ms  {pairs / ms * 1e3:.3e} pairs/s")

    add("softmin_bwd_x_p2", lambda: hip.softmin_bwd_x_raw(x, y, h, out, g, eps, 2))
    add("gaussian_product", lambda: hip.kernel_conv_fwd_raw(hip.GAUSSIAN, x, y, v, blur))
    add("gaussian_gradient", lambda: hip.kernel_conv_bwd_x_raw(hip.GAUSSIAN, x, y, v, g, blur))
    add("gaussian_product_and_gradient", lambda: hip.kernel_conv_fwd_grad_raw(hip.GAUSSIAN, x, y, v, blur))
    # distance-type reductions, RAW C-ABI calls: a launch this big sorts its clouds inside the library (csrc/glhip_autosort.h, round 6:
    # both voxel sorts are inside the timed call) and runs the matrix-core distance kernel; GLHIP_FLAG_NO_SORT: the generic kernel
    add("softmin_fwd_p1", lambda: hip.softmin_fwd_raw(x, y, h, 0.05, 1), reps=2)
    add("laplacian_product", lambda: hip.kernel_conv_fwd_raw(hip.LAPLACIAN, x, y, v, blur), reps=2)
    add("energy_product", lambda: hip.kernel_conv_fwd_raw(hip.ENERGY, x, y, v, blur), reps=2)
    add("softmin_fwd_p1_direct_differences", lambda: hip.softmin_fwd_raw(x, y, h, 0.05, 1, flags=hip.FLAG_NO_SORT), reps=1)
    # 4 <= D <= 16 (csrc/glhip_softmin_xd.h): f16 x 2 exponents, ceil((3 D + 6) / 16) chained MFMAs (GLHIP_FLAG_F16X2, in range on the
    # unit cube), and the default bf16 x 3 layout, ceil(6 (D + 1) / 16)
    gd = torch.Generator().manual_seed(11)
    for D in (4, 5, 8, 12, 16):
        xd = torch.rand(1, n, D, generator=gd).to(dev)
        yd = torch.rand(1, n, D, generator=gd).to(dev)
        add(f"softmin_fwd_p2_d{D}", lambda: hip.softmin_fwd_raw(xd, yd, h, eps, 2, flags=HEADLINE_FLAGS), reps=2)
        if D in (8, 16):
            add(f"softmin_fwd_p2_d{D}_bf16x3", lambda: hip.softmin_fwd_raw(xd, yd, h, eps, 2), reps=1)
            add(f"softmin_bwd_x_p2_d{D}", lambda: hip.softmin_bwd_x_raw(xd, yd, h, out, g, eps, 2, flags=HEADLINE_FLAGS), reps=1)
        if D == 4:
            add("gaussian_product_d4", lambda: hip.kernel_conv_fwd_raw(hip.GAUSSIAN, xd, yd, v, 2 * blur), reps=2)
    # float64 clouds (csrc/glhip_api_f64.hip): no matrix cores, inlined exp — timed on a tenth of the rows
    m = max(n // 10, 1)
    x64, y64, h64 = x[:, :m].double().contiguous(), y.double(), h.double()
    ms = event_ms(lambda: hip.softmin_fwd_raw(x64, y64, h64, eps, 2), 1)
    res["softmin_fwd_p2_f64"] = {"ms": ms, "rows": m, "pairs_per_s": float(m) * n / (ms * 1e-3)}
    log(f"[bench] softmin_fwd_p2_f64 ({m} rows x {n} columns): {ms:.2f} ms  {float(m) * n / ms * 1e3:.3e} pairs/s")
    return res


def sinkhorn_pairs(n_eps, B, N, M, debias=True):
    """Soft-min pair evaluations of one loss: (1 + n_eps + 1) rounds of the 4 (2 without debias) simultaneous reductions."""
    per_round = (2 * N * M + N * N + M * M) if debias else 2 * N * M
    return float(B) * (n_eps + 2) * per_round


def cfg4_batch(dev, B, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 4096, 3, generator=g).to(dev).bfloat16()
    y = torch.rand(B, 4096, 3, generator=g).to(dev).bfloat16()
    return x, y


CFG4 = dict(p=2, blur=0.05, diameter=1.8, scaling=0.5)


def cfg4_pairs(B):
    from geomloss_amd.sinkhorn_divergence import epsilon_schedule
    return sinkhorn_pairs(len(epsilon_schedule(2, CFG4["diameter"], CFG4["blur"], CFG4["scaling"])), B, 4096, 4096)


def sinkhorn_wallclock(dev):
    """End-to-end SamplesLoss timings on the BASELINE configs (single GPU, outside the timed region)."""
    from geomloss_amd import SamplesLoss

    out = {}

    def run(name, loss, n, backward, reps=2):
        g = torch.Generator().manual_seed(1)
        x = torch.rand(n, 3, generator=g).to(dev).requires_grad_(backward)
        y = torch.rand(n, 3, generator=g).to(dev)
        ts = []
        for _ in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            L = loss(x, y)
            if backward:
                torch.autograd.grad(L, [x])
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        out[name] = {"seconds": min(ts[1:]), "first_call_seconds": ts[0], "loss": float(L.detach())}
        log(f"[bench] {name}: {min(ts[1:]):.4f} s (first call {ts[0]:.3f} s)")

    run("multiscale_1e6_fwd", SamplesLoss("sinkhorn", p=2, blur=0.05, backend="multiscale"), 1_000_000, False)
    run("multiscale_1e6_fwd_bwd", SamplesLoss("sinkhorn", p=2, blur=0.05, backend="multiscale"), 1_000_000, True)
    # the sizes from which the reference's backend="auto" picks the two-scale solver (N M > 1e8): clusters of a few points
    run("multiscale_1e4_fwd", SamplesLoss("sinkhorn", p=2, blur=0.05, backend="multiscale"), 10_000, False, reps=4)
    run("multiscale_1e5_fwd", SamplesLoss("sinkhorn", p=2, blur=0.05, backend="multiscale"), 100_000, False, reps=3)
    run("online_1e5_fwd", SamplesLoss("sinkhorn", p=2, blur=0.05, backend="online"), 100_000, False)
    run("online_1e5_fwd_bwd", SamplesLoss("sinkhorn", p=2, blur=0.05, backend="online"), 100_000, True)
    run("gaussian_online_1e6_fwd", SamplesLoss("gaussian", blur=0.05, backend="online"), 1_000_000, False, reps=1)
    run("gaussian_online_1e6_fwd_bwd", SamplesLoss("gaussian", blur=0.05, backend="online"), 1_000_000, True, reps=1)
    run("energy_online_1e6_fwd", SamplesLoss("energy", backend="online"), 1_000_000, False, reps=1)
    run("energy_online_1e6_fwd_bwd", SamplesLoss("energy", backend="online"), 1_000_000, True, reps=1)
    run("gaussian_multiscale_1e6_fwd", SamplesLoss("gaussian", blur=0.05, backend="multiscale"), 1_000_000, False, reps=1)

    # the reference's recipe in dimension > 3 (examples/sinkhorn_multiscale/plot_optimal_transport_cluster.py:155-166): clusters given
    # as labels — voxels of the 3 spatial coordinates of (position, feature) points — and the two-scale solver on the 4-D clouds
    def labels4d(t, scale=0.08):
        q = (t[:, :3] / scale).floor().long()
        return torch.unique((q[:, 0] * 64 + q[:, 1]) * 64 + q[:, 2], return_inverse=True)[1].int()     # compact 0..C-1, like grid_cluster

    g = torch.Generator().manual_seed(3)
    x4, y4 = torch.rand(200_000, 4, generator=g).to(dev), torch.rand(200_000, 4, generator=g).to(dev)
    loss4 = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="multiscale")
    w4 = torch.full((200_000,), 1.0 / 200_000, device=dev)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        L = loss4(labels4d(x4), w4, x4, labels4d(y4), w4, y4)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    out["multiscale_4d_labels_2e5_fwd"] = {"seconds": min(ts[1:]), "first_call_seconds": ts[0], "loss": float(L)}
    log(f"[bench] multiscale_4d_labels_2e5_fwd: {min(ts[1:]):.4f} s")
    return out


def sharded_reference(dev, B=256, reps=3):
    """BASELINE configs[3] with the whole batch on ONE GPU: the single-GPU point of the `--gpus N` curve."""
    from geomloss_amd import SamplesLoss

    x, y = cfg4_batch(dev, B, seed=2)
    loss = SamplesLoss("sinkhorn", backend="online", **CFG4)
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        L = loss(x, y).sum()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    return {"workload": f"SamplesLoss('sinkhorn', online) B={B} N=M=4096 3D bf16, diameter=1.8, forward", "seconds": t,
            "pairs_per_s": cfg4_pairs(B) / t, "loss_sum": float(L)}


def settle_host():
    """One full Python garbage collection now, survivors moved out of the collector's reach (``gc.freeze``).  Why this is part of
    the protocol: the first generation-2 collection of a process that has imported torch walks ~1e6 objects and takes 30-40 ms
    (measured: profiles/r05_shard_stall.txt); it fires once, a few thousand allocations in — inside the timed region or not, by luck —
    and while the host is stopped the launch queue of a 2 ms step drains.  That was the "B = 32 anomaly" of the round-4 review."""
    import gc
    gc.collect()
    gc.freeze()


def shard_curve(dev, sizes=(256, 128, 64, 32), calls=200):
    """What one rank of `--gpus W` computes per step, for W = 256 / B, measured on THIS GPU: B problems of BASELINE configs[3],
    every call bracketed by HIP events on the launch stream.  median / p99 / max per call + the share of the ideal (B / 256 of the
    B = 256 median) — the single-GPU evidence behind the 1 -> 8 curve (no collective is on the data path)."""
    from geomloss_amd import SamplesLoss

    loss = SamplesLoss("sinkhorn", backend="online", **CFG4)
    out = {"workload": "SamplesLoss('sinkhorn', online) forward, B x 4096 x 4096 3D bf16, diameter=1.8; per-call HIP-event times", "calls": calls,
           "shards": {}}
    for B in sizes:
        x, y = cfg4_batch(dev, B, seed=2)
        for _ in range(5):
            loss(x, y).sum()
        settle_host()
        torch.cuda.synchronize()
        a = [torch.cuda.Event(enable_timing=True) for _ in range(calls)]
        b = [torch.cuda.Event(enable_timing=True) for _ in range(calls)]
        t0 = time.perf_counter()
        for k in range(calls):
            a[k].record()
            loss(x, y).sum()
            b[k].record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / calls * 1e3
        ms = sorted(s.elapsed_time(e) for s, e in zip(a, b))
        med = ms[len(ms) // 2]
        out["shards"][str(B)] = {"gpus_equivalent": 256 / B, "median_ms": med, "p99_ms": ms[min(calls - 1, int(0.99 * calls))], "max_ms": ms[-1],
                                 "wall_ms_per_call": wall, "stalls_over_1p3x_median": sum(1 for t in ms if t > 1.3 * med),
                                 "pairs_per_s_at_median": cfg4_pairs(B) / (med * 1e-3)}
        log(f"[bench] shard_curve B={B}: median {med:.3f} ms, p99 {out['shards'][str(B)]['p99_ms']:.3f}, max {ms[-1]:.3f}, wall {wall:.3f} ms/call")
        if B == sizes[-1]:
            # round-5 advice: the same calls with the collector left as a fresh process has it (no settle_host): how many stall, how long
            import gc
            gc.unfreeze()
            torch.cuda.synchronize()
            for k in range(calls):
                a[k].record()
                loss(x, y).sum()
                b[k].record()
            torch.cuda.synchronize()
            raw = sorted(s.elapsed_time(e) for s, e in zip(a, b))
            out["shards"][str(B)]["without_settle_host"] = {"median_ms": raw[len(raw) // 2], "max_ms": raw[-1],
                                                            "stalls_over_1p3x_median": sum(1 for t in raw if t > 1.3 * raw[len(raw) // 2])}
            settle_host()
    if "256" in out["shards"]:
        full = out["shards"]["256"]["median_ms"]
        for B in sizes:
            r = out["shards"][str(B)]
            r["share_of_ideal"] = full * B / 256 / r["median_ms"]
    return out


def run_headline(args, dev):
    from geomloss_amd import hip

    n = args.points
    x, y, h, eps = make_problem(n, dev, seed=1000)
    for _ in range(args.warmup):
        hip.softmin_fwd_raw(x, y, h, eps, 2, flags=HEADLINE_FLAGS)
    settle_host()
    torch.cuda.synchronize()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        starts[k].record()               # same stream as the launch (torch's current stream)
        hip.softmin_fwd_raw(x, y, h, eps, 2, flags=HEADLINE_FLAGS)
        stops[k].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in zip(starts, stops)) / args.steps

    pairs_per_launch = float(n) * n
    value = pairs_per_launch * args.steps / elapsed
    kernel_pairs_s = pairs_per_launch / (kernel_ms * 1e-3)
    dense_gbs = kernel_pairs_s * 4 / 1e9
    compulsory = 4.0 * (n * 3 + n * 4 + n)
    builder_pmc = None
    pmc = os.path.join(ROOT, PMC_SUMMARY)
    if os.path.exists(pmc):
        try:
            builder_pmc = dict(json.load(open(pmc)), source=f"{PMC_SUMMARY}: rocprofv3 --pmc passes of this command, run by the builder "
                                                            "(tools/profile_gpu.sh) — NOT measured in this process")
        except Exception:
            builder_pmc = None
    res = {
        "metric": "softmin pairs/s, N=M=1e6 3D fp32 (Sinkhorn wall-clock: `sinkhorn_wallclock`; roofline: `roofline`)"
                  if n == 1_000_000 else f"softmin pairs/s (N=M={n} 3D fp32)",
        "value": value, "unit": "pairs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"glhip_softmin_fwd dense, N=M={n}, D=3, p=2, eps=0.05^2, uniform unit-cube clouds; the reduction "
                        "behind SamplesLoss('sinkhorn', backend='online'/'multiscale') (BASELINE configs[1]-[2]); one problem, one GPU",
            "pairs_per_step": pairs_per_launch,
            "parallelism": "single GPU",
            "exponent_layout": "GLHIP_FLAG_F16X2, range vouched by the caller (diameter^2 / eps = 1200 here; the flag's contract allows ~1.5e5) — the "
                               "launch SamplesLoss('sinkhorn', blur=.05) itself makes on these clouds; the default of a raw C-ABI call (bf16 x 3, no "
                               "vouching needed) is timed in the same run: roofline.bf16x3_layout",
        },
        "roofline": {
            "bound": "valu", "achieved": kernel_pairs_s / 1e12, "peak": VALU_PEAK_PAIRS_PER_S / 1e12, "unit": "Tpair/s",
            "frac": kernel_pairs_s / VALU_PEAK_PAIRS_PER_S, "traffic": None,
            "model": "VALU issue: 1 v_exp_f32 (quarter rate, 8 cycles per wave64) + 1 v_add_f32 (2 cycles) per pair, the one f16 x 2 "
                     "MFMA per 1024 pairs that forms the exponents co-issues -> 10 SIMD cycles per 64 pairs; 256 CU x 4 SIMD x 2.4 GHz",
            "kernel": "xd_fwd_kernel<XD_SOFTMIN, D = 3, f16 x 2> (+ xd_pack_kernel + merge_kernel: one glhip_softmin_fwd call with "
                      "GLHIP_FLAG_F16X2; the reducing kernel is > 99.9 % of it)",
            "kernel_ms": kernel_ms, "kernel_pairs_per_s": kernel_pairs_s,
            "frac_of_measured_stream": kernel_pairs_s / (SIMDS * CLOCK_HZ * 64 / MEASURED_STREAM_CYCLES),
            "measured_stream": f"{MEASURED_STREAM_CYCLES} cycles per 64 pairs: the same exp2 + add instruction stream micro-benchmarked "
                               "alone on this part (profiles/r01_ubench_pipes.txt)",
            "hbm_dense_equivalent": {
                "achieved": dense_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dense_gbs / HBM_PEAK_GBS,
                "model": "BASELINE.json's figure (SURVEY §8d-1): 4 algorithmic bytes per pair = the fp32 cost entry the tensorized "
                         "formulation streams; > 1 because this kernel never materialises the matrix — not a physical HBM rate",
            },
            "compulsory_bytes_per_launch": compulsory, "compulsory_GBs": compulsory / (kernel_ms * 1e-3) / 1e9,
            "builder_pmc": builder_pmc,
        },
    }
    if not args.no_extras:
        ms3 = event_ms(lambda: hip.softmin_fwd_raw(x, y, h, eps, 2), 3)
        res["roofline"]["bf16x3_layout"] = {"kernel_ms": ms3, "kernel_pairs_per_s": pairs_per_launch / (ms3 * 1e-3),
                                            "note": "the same launch without GLHIP_FLAG_F16X2: softmin_fwd_x32_kernel, three bf16 pieces per "
                                                    "operand, two MFMAs per 1024 pairs — what a raw C-ABI call gets when it does not vouch for the range"}
        log(f"[bench] same launch, bf16 x 3 layout: {ms3:.2f} ms")
        if not args.no_traffic:
            t = measure_traffic(n)
            if t is not None:
                res["roofline"]["traffic"] = t["bytes_per_launch"]
                res["roofline"]["traffic_detail"] = t
                log(f"[bench] HBM traffic of the dominant kernel (PMC, this run): {t['bytes_per_launch'] / 1e6:.0f} MB per launch")
        for key, fn in (("kernels", lambda: hot_path_kernels(dev, n)), ("cpu_baseline", cpu_baseline),
                        ("sinkhorn_wallclock", lambda: sinkhorn_wallclock(dev)),
                        ("sharded_batch_reference", lambda: sharded_reference(dev)), ("shard_curve", lambda: shard_curve(dev))):
            try:
                res[key] = fn()
            except Exception as e:   # never lose the GPU number to a side leg
                res[key] = {"error": repr(e)}
                if key == "cpu_baseline":
                    res[key].update(value=None, unit="pairs/s", cores=os.cpu_count(), kind="port", sample="failed")
    emit(res)


def run_sharded(args, dev, rank, world):
    """BASELINE configs[3]: B = 256 problems sharded over the ranks through ShardedSamplesLoss."""
    import torch.distributed as dist
    from geomloss_amd import SamplesLoss
    from geomloss_amd.distributed import ShardedSamplesLoss, shard_bounds

    B = args.batch
    lo, hi = shard_bounds(B, rank, world)
    # every rank draws the whole batch from the same seed and keeps its slice: the global problem does not depend on N
    x, y = cfg4_batch(torch.device("cpu"), B, seed=2)
    x, y = x[lo:hi].to(dev), y[lo:hi].to(dev)
    loss = ShardedSamplesLoss(SamplesLoss("sinkhorn", backend="online", **CFG4), reduction="sum")

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss(x, y)
    settle_host()          # the process's one 30-40 ms full garbage collection happens here, not inside a 2 ms step
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        total = loss(x, y)          # local batched loss + scalar all-reduce (RCCL over xGMI)
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    devs = [None] * world                      # which device every rank really computed on: the curve's n_gpus must be real GPUs
    dist.all_gather_object(devs, f"{os.uname().nodename}:cuda{dev.index}")
    # Attribution, OUTSIDE the timed region: what a sub-linear point of the curve is made of.  (a) every rank's own shard without
    # any collective (its kernels + launches), (b) the scalar all-reduce alone; the rest of ms_per_step is waiting for the slowest rank.
    reps = max(2, min(args.steps, 5))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(reps):
        local = loss.loss(x, y).sum()
    torch.cuda.synchronize()
    local_ms = (time.perf_counter() - t1) / reps * 1e3
    scalar = local.detach().clone()
    fence()
    t2 = time.perf_counter()
    for _ in range(20):
        dist.all_reduce(scalar)
    torch.cuda.synchronize()
    allreduce_ms = (time.perf_counter() - t2) / 20 * 1e3
    per_rank = [None] * world
    dist.all_gather_object(per_rank, {"rank": rank, "problems": hi - lo, "local_loss_ms": round(local_ms, 4), "allreduce_ms": round(allreduce_ms, 4)})
    if rank == 0:
        pairs = cfg4_pairs(B)
        emit(({
            "metric": "softmin pairs/s of the batch-sharded Sinkhorn loss (BASELINE configs[3]: B=256, N=M=4096 3D bf16)",
            "value": pairs * args.steps / elapsed, "unit": "pairs/s", "n_gpus": world, "ranks_seen": dist.get_world_size(),
            "backend": dist.get_backend(), "devices": sorted(set(devs)), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "bf16 points, f32 dual variables and accumulation", "data": "synthetic",
            "config": {
                "workload": f"ShardedSamplesLoss(SamplesLoss('sinkhorn', p=2, blur=.05, diameter=1.8, backend='online')), B={B} x 4096 x 4096 "
                            f"3D bf16, {hi - lo} problems on rank 0; one step = one forward loss of the whole batch",
                "pairs_per_step": pairs, "global_batch": B,
                "parallelism": f"batch sharded x{world} (contiguous slices), no data-path collective, one scalar all-reduce per step "
                               f"({args.backend})",
                "single_gpu_point": "`sharded_batch_reference` of the N=1 line (same workload, whole batch on one GPU); per-shard "
                                    "timings of one GPU: `shard_curve` of the same line",
                "protocol": "W warm-up steps, one full Python garbage collection + gc.freeze (settle_host), barrier, K timed steps, barrier",
            },
            "loss_sum": float(total),
            "per_rank_ms": [r["local_loss_ms"] for r in sorted(per_rank, key=lambda r: r["rank"])],
            "allreduce_ms": max(r["allreduce_ms"] for r in per_rank),
            "attribution": {
                "note": "measured after the timed region: per_rank_ms = each rank's shard alone, no collective (kernels + launches); "
                        "allreduce_ms = one scalar all-reduce (slowest rank, 20 back to back); ms_per_step - max(per_rank_ms) - "
                        "allreduce_ms = waiting on the slowest rank + the barrier of the protocol",
                "problems_per_rank": [r["problems"] for r in sorted(per_rank, key=lambda r: r["rank"])],
            },
        }))


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks here — one process per GPU through
    `torch.distributed.run`, rendezvous on 127.0.0.1 — and hand rank 0's JSON line through.  Refuses (non-zero exit code,
    nothing on stdout) instead of falling back to the 1-GPU headline when the node has fewer than N GPUs."""
    import socket
    import subprocess

    if not (args.single_device or args.dry_run):
        have = torch.cuda.device_count()
        if have < args.gpus:
            log(f"[bench] refused: --gpus {args.gpus} but this node exposes {have} GPU(s) "
                "(use --single-device --backend gloo for a dry run of the sharded path on one GPU)")
            return 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    log(f"[bench] WORLD_SIZE unset: launching {args.gpus} ranks: {' '.join(cmd)}")
    return subprocess.run(cmd, env=env).returncode


def run_dry(args, rank, world):
    """`--dry-run`: the launch / rendezvous / barrier / all-reduce / max-over-ranks plumbing of the N > 1 leg with a stand-in
    CPU workload and the gloo backend — what a box without GPUs can check.  The line says so and carries no throughput."""
    import torch.distributed as dist

    from geomloss_amd.distributed import shard_bounds
    lo, hi = shard_bounds(args.batch, rank, world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        total = torch.tensor([float(hi - lo)])
        dist.all_reduce(total)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        emit(({"metric": "dry run of the batch-sharded leg (no kernels timed)", "value": None, "unit": "pairs/s",
                          "n_gpus": world, "ranks_seen": dist.get_world_size(), "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": t.item() / args.steps * 1e3, "dry_run": True, "backend": "gloo",
                          "items_all_ranks": float(total)}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--points", type=int, default=1_000_000, help="N = M of the soft-min workload (N = 1)")
    ap.add_argument("--batch", type=int, default=256, help="global batch of the sharded workload (N > 1)")
    ap.add_argument("--no-extras", action="store_true", help="skip the side legs (traffic, kernels, cpu_baseline, wall-clocks)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 PMC passes that measure roofline.traffic")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for a 1-GPU dry run)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the N>1 (batch-sharded) leg even with one rank: exercises process-group init + the RCCL all-reduce on a 1-GPU box")
    ap.add_argument("--single-device", action="store_true",
                    help="dry run of the N>1 path on a 1-GPU box: every rank uses cuda:0 (only with --backend gloo)")
    ap.add_argument("--dry-run", action="store_true",
                    help="N>1 plumbing only (spawn, rendezvous, collectives) on the CPU with gloo: no GPU, no kernels, no throughput")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.single_device and args.backend == "nccl":
        ap.error("--single-device puts every rank on cuda:0, which RCCL refuses: add --backend gloo")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))      # not under torchrun: start the ranks ourselves
    claim_stdout()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        log(f"[bench] refused: launched with WORLD_SIZE={world} but --gpus {args.gpus}; the two must agree")
        sys.exit(2)
    if args.dry_run:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            run_dry(args, rank, world)
        finally:
            dist.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if args.single_device:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        log(f"[bench] refused: rank {rank} wants cuda:{local_rank} but the node exposes {torch.cuda.device_count()} GPU(s)")
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from geomloss_amd import hip
    hip.load_library()   # raises if the HIP extension is missing: there is no fallback to time

    if world == 1 and not args.force_sharded:
        run_headline(args, dev)
        return
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    if args.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    try:
        run_sharded(args, dev, rank, world)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
