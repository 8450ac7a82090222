"""Size-independent properties of the drop-in loss on the GPU (HIP backends)."""

import pytest
import torch

from geomloss_amd import SamplesLoss, hip

pytestmark = pytest.mark.gpu


def _clouds(cuda, seed, N, M, D=3):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(N, D, generator=g).to(cuda)
    y = (torch.rand(M, D, generator=g) * 0.7 + 0.2).to(cuda)
    a = torch.rand(N, generator=g).to(cuda) + 0.1
    b = torch.rand(M, generator=g).to(cuda) + 0.1
    return a / a.sum(), x, b / b.sum(), y


@pytest.mark.parametrize("loss,kw", [("sinkhorn", dict(p=2, blur=0.05)), ("sinkhorn", dict(p=1, blur=0.1)),
                                      ("gaussian", dict(blur=0.1)), ("energy", dict())])
def test_symmetry_and_invariances(cuda, loss, kw):
    a, x, b, y = _clouds(cuda, 0, 1500, 1700)
    L = SamplesLoss(loss, backend="online", diameter=2.0, **kw)
    v = L(a, x, b, y).item()
    assert abs(L(b, y, a, x).item() - v) < 1e-5 * abs(v)                       # S(a,b) = S(b,a)
    perm = torch.randperm(1500, generator=torch.Generator().manual_seed(1)).to(cuda)
    assert abs(L(a[perm], x[perm], b, y).item() - v) < 1e-5 * abs(v)           # relabelling the samples
    shift = torch.tensor([3.0, -2.0, 1.0], device=cuda)
    assert abs(L(a, x + shift, b, y + shift).item() - v) < 2e-4 * abs(v)       # translation (inputs re-rounded)
    assert v > 0


def test_loss_of_a_measure_with_itself_vanishes_with_its_gradient(cuda):
    a, x, _, _ = _clouds(cuda, 2, 2000, 10)
    x = x.clone().requires_grad_(True)
    L = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="online")(a, x, a, x.detach().clone())
    (g,) = torch.autograd.grad(L, [x])
    assert abs(L.item()) < 1e-6 and g.abs().max().item() < 1e-6


def test_batched_loss_equals_per_item_losses(cuda):
    g = torch.Generator().manual_seed(3)
    B, N, M = 4, 600, 700
    x = torch.rand(B, N, 3, generator=g).to(cuda).requires_grad_(True)
    y = (torch.rand(B, M, 3, generator=g) * 0.5 + 0.4).to(cuda)
    L = SamplesLoss("sinkhorn", p=2, blur=0.05, diameter=1.8, backend="online")
    vb = L(x, y)
    (gb,) = torch.autograd.grad(vb.sum(), [x])
    for k in range(B):
        xk = x[k].detach().clone().requires_grad_(True)
        vk = L(xk, y[k])
        (gk,) = torch.autograd.grad(vk, [xk])
        assert abs(vk.item() - vb[k].item()) < 1e-6 * abs(vk.item())
        assert (gk - gb[k]).abs().max().item() < 1e-6 * gk.abs().max().item() + 1e-9


def test_online_and_multiscale_agree_when_the_jump_is_last(cuda):
    """With diameter = 1 and a tiny cluster scale the coarse level only extrapolates: both backends then solve the
    same fine problem up to the coarse initialisation."""
    a, x, b, y = _clouds(cuda, 6, 4000, 4200)
    x, y = x * 0.5, y * 0.5
    kw = dict(p=2, blur=0.05, diameter=1.0, scaling=0.8)
    Lo = SamplesLoss("sinkhorn", backend="online", **kw)(a, x, b, y).item()
    Lm = SamplesLoss("sinkhorn", backend="multiscale", cluster_scale=0.01, **kw)(a, x, b, y).item()
    assert abs(Lo - Lm) < 5e-3 * abs(Lo)


def test_sharded_loss_on_gpu_single_rank(cuda):
    """ShardedSamplesLoss over the HIP backend (world size 1, gloo): same value / gradient as the plain loss and the
    global-diameter collective path (diameter=None) runs on GPU tensors."""
    import os
    import torch.distributed as dist
    from geomloss_amd.distributed import ShardedSamplesLoss

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        g = torch.Generator().manual_seed(9)
        x = torch.rand(3, 500, 3, generator=g).to(cuda).bfloat16().requires_grad_(True)
        y = (torch.rand(3, 600, 3, generator=g) * 0.6 + 0.3).to(cuda).bfloat16()
        base = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="online")
        ref = base(x, y)
        tot = ShardedSamplesLoss(base, "sum")(x, y)
        vec = ShardedSamplesLoss(base, "none")(x, y)
        assert abs(tot.item() - ref.sum().item()) < 1e-6 * abs(tot.item())
        assert (vec - ref).abs().max().item() < 1e-7
        (g1,) = torch.autograd.grad(tot, [x])
        assert torch.isfinite(g1.float()).all() and g1.float().abs().max() > 0
    finally:
        dist.destroy_process_group()


def test_hipgraph_mode_reproduces_the_eager_loop(cuda):
    """GEOMLOSS_HIP_GRAPH / set_graph_mode: the captured annealing loop gives the same loss, gradient and potentials,
    also when replayed on new data of the same shape."""
    from geomloss_amd import sinkhorn_samples as ss

    L = SamplesLoss("sinkhorn", p=2, blur=0.05, diameter=1.8, backend="online")
    Lp = SamplesLoss("sinkhorn", p=2, blur=0.05, diameter=1.8, backend="online", potentials=True)
    results = {}
    for mode in (False, True, True):   # eager, capture, replay
        ss.set_graph_mode(mode)
        try:
            out = []
            for seed in (0, 1):
                a, x, b, y = _clouds(cuda, seed, 900, 1100)
                x = x.clone().requires_grad_(True)
                v = L(a, x, b, y)
                (g,) = torch.autograd.grad(v, [x])
                F, G = Lp(a, x.detach(), b, y)
                out.append((v.item(), g, F))
            results.setdefault(mode, []).append(out)
        finally:
            ss.set_graph_mode(False)
    eager = results[False][0]
    for run in results[True]:
        for (v0, g0, F0), (v1, g1, F1) in zip(eager, run):
            assert abs(v0 - v1) <= 1e-7 * abs(v0)
            assert (g0 - g1).abs().max().item() <= 1e-7 * g0.abs().max().item()
            assert (F0 - F1).abs().max().item() <= 1e-7


def test_hipgraph_replay_with_converted_inputs(cuda):
    """Graph mode with inputs the kernels cannot read as they are (fp16: widened to fp32; transposed storage: made
    contiguous).  The conversions must be part of the captured work: a replay on NEW data of the same shape has to see the new
    data, and nothing the graph touches may be freed between replays."""
    from geomloss_amd import sinkhorn_samples as ss

    L = SamplesLoss("sinkhorn", p=2, blur=0.05, diameter=1.8, backend="online")

    def inputs(seed, kind):
        g = torch.Generator().manual_seed(seed)
        x, y = torch.rand(800, 3, generator=g).to(cuda), (torch.rand(900, 3, generator=g) * 0.8 + 0.1).to(cuda)
        if kind == "fp16":
            return x.half(), y.half()
        return x.t().contiguous().t(), y.t().contiguous().t()     # (N,3) views of (3,N) storage

    for kind in ("fp16", "transposed"):
        eager = []
        for seed in (0, 1, 2):
            x, y = inputs(seed, kind)
            eager.append(L(x, y).item())
        assert abs(eager[0] - eager[1]) > 1e-3 * abs(eager[0])     # the three problems really differ
        ss.set_graph_mode(True)
        try:
            graphed = []
            for seed in (0, 1, 2):                                # capture, replay, replay
                x, y = inputs(seed, kind)
                graphed.append(L(x, y).item())
                torch.cuda.empty_cache()                          # would release a plan that lived outside the graph's pool
                junk = torch.randn(1 << 20, device=cuda)          # ... and this would be handed its memory
                del junk
        finally:
            ss.set_graph_mode(False)
        for e, g_ in zip(eager, graphed):
            assert abs(e - g_) <= 1e-6 * abs(e), (kind, eager, graphed)


@pytest.mark.parametrize("kw", [dict(), dict(debias=False), dict(reach=0.5), dict(potentials=True)])
def test_fused_iterations_reproduce_the_per_softmin_loop(cuda, kw):
    """One launch per iteration (glhip_sinkhorn_iter4 + the fused last step) vs four glhip_sinkhorn_step launches and four
    autograd soft-mins: same losses, potentials and gradients, batched and un-batched, fp32 and bf16 points."""
    from geomloss_amd import sinkhorn_samples as ss

    L = SamplesLoss("sinkhorn", p=2, blur=0.05, scaling=0.6, backend="online", **kw)
    for B, dtype in ((None, torch.float32), (3, torch.bfloat16)):
        torch.manual_seed(5)
        shp = (lambda n: (n, 3)) if B is None else (lambda n: (B, n, 3))
        x = torch.rand(shp(600), device=cuda).to(dtype).requires_grad_(True)
        y = torch.rand(shp(700), device=cuda).to(dtype)
        res = {}
        for fused in (True, False):
            ss.set_iteration_fusion(fused)
            try:
                out = L(x, y)
                v = out[0].sum() + out[1].sum() if kw.get("potentials") else out.sum()
                (g,) = torch.autograd.grad(v, [x])
                res[fused] = (v.item(), g.float())
            finally:
                ss.set_iteration_fusion(True)
        (v1, g1), (v0, g0) = res[True], res[False]
        assert abs(v1 - v0) <= 2e-6 * abs(v0) + 1e-9
        # bf16 points get bf16 gradients: the two paths may round a value to neighbouring bf16 numbers (1 ulp = 2^-8)
        gtol = 2e-5 if dtype == torch.float32 else 2 ** -7
        assert (g1 - g0).abs().max().item() <= gtol * g0.abs().max().item() + 1e-9


@pytest.mark.parametrize("backend,D,p,kw", [("online", 3, 2, dict()), ("online", 3, 2, dict(debias=False, reach=0.5)), ("online", 3, 1, dict()),
                                            ("online", 6, 2, dict(potentials=True)), ("multiscale", 3, 2, dict()),
                                            ("multiscale", 2, 2, dict(reach=0.7))])
def test_library_side_annealing_is_the_same_loop(cuda, backend, D, p, kw):
    """glhip_sinkhorn_anneal (every iteration of a level queued by one library call) against one glhip_sinkhorn_iter4 call per
    temperature from Python: the same launches with the same arguments — identical losses, potentials and gradients, bit for bit;
    batched bf16 points and a given diameter included."""
    from geomloss_amd import sinkhorn_samples as ss

    for B, dtype, diameter in ((None, torch.float32, None), (3, torch.bfloat16, 2.0)):
        if backend == "multiscale" and B is not None:
            continue
        L = SamplesLoss("sinkhorn", p=p, blur=0.05, scaling=0.6, backend=backend, diameter=diameter, **kw)
        torch.manual_seed(6)
        shp = (lambda n: (n, D)) if B is None else (lambda n: (B, n, D))
        x = torch.rand(shp(900), device=cuda).to(dtype).requires_grad_(True)
        y = torch.rand(shp(700), device=cuda).to(dtype)
        res = {}
        for on in (True, False):
            ss._anneal_in_library = on
            try:
                out = L(x, y)
                v = out[0].sum() + out[1].sum() if kw.get("potentials") else out.sum()
                (g,) = torch.autograd.grad(v, [x])
                res[on] = (v.detach().clone(), g.clone())
            finally:
                ss._anneal_in_library = True
        assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])


def test_bench_sharded_path_two_ranks_on_one_gpu(cuda):
    """`bench.py --gpus 2` — the N > 1 leg the driver launches on a multi-GPU node (BASELINE configs[3] through
    ShardedSamplesLoss) — exercised here with two processes sharing cuda:0 over gloo; its loss must equal the unsharded one."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "6",
           "--backend", "gloo", "--single-device"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["value"] > 0 and rec["config"]["global_batch"] == 6
    sys.path.insert(0, root)
    import bench
    x, y = bench.cfg4_batch(cuda, 6, seed=2)
    ref = SamplesLoss("sinkhorn", backend="online", **bench.CFG4)(x, y).sum().item()
    assert abs(rec["loss_sum"] - ref) <= 1e-6 * abs(ref)
    assert abs(rec["config"]["pairs_per_step"] - 6 * 40 * 4096.0**2) < 1
    # attribution of a sub-linear point (measured after the timed region): one local time per rank, the bare all-reduce
    assert len(rec["per_rank_ms"]) == 2 and all(t > 0 for t in rec["per_rank_ms"]) and rec["allreduce_ms"] > 0
    assert rec["attribution"]["problems_per_rank"] == [3, 3]


def test_bench_gpus2_without_torchrun(cuda):
    """`python bench.py --gpus 2 --backend gloo --single-device` with no launcher around it: bench.py starts the two ranks itself
    (both on cuda:0 here) and reports n_gpus = 2 with the loss of the unsharded batch."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "4",
           "--backend", "gloo", "--single-device"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["backend"] == "gloo" and rec["value"] > 0
    assert len(rec["devices"]) == 1                      # --single-device: the line says both ranks shared one GPU
    sys.path.insert(0, root)
    import bench
    x, y = bench.cfg4_batch(cuda, 4, seed=2)
    ref = SamplesLoss("sinkhorn", backend="online", **bench.CFG4)(x, y).sum().item()
    assert abs(rec["loss_sum"] - ref) <= 1e-6 * abs(ref)


def test_bench_sharded_path_on_rccl_single_rank(cuda):
    """The same leg over the real backend (`nccl` = RCCL): one rank, so that process-group creation, the barrier and the scalar
    all-reduce run through RCCL on this GPU (two ranks cannot share a device under RCCL)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-sharded", "--steps", "2", "--warmup", "1",
           "--batch", "4", "--backend", "nccl"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and "nccl" in rec["config"]["parallelism"]


@pytest.mark.parametrize("backend,kw", [("online", dict()), ("online", dict(debias=False)), ("online", dict(reach=0.5)),
                                         ("multiscale", dict(scaling=0.7)), ("multiscale", dict(scaling=0.7, debias=False))])
def test_one_reduction_final_update_equals_the_two_pass_one(cuda, monkeypatch, backend, kw):
    """The last, differentiable update of the loop as ONE reduction (value + gradient from the previous iterate as a bound,
    glhip_softmin_fwd_grad) against the forward + backward pair it replaces on big launches; forced on at a small size here."""
    from geomloss_amd import sinkhorn_samples as ss
    g = torch.Generator().manual_seed(31)
    shapes = [(None, torch.float32)] + ([(3, torch.bfloat16)] if backend == "online" else [])
    for B, dtype in shapes:
        shp = (lambda n: (n, 3)) if B is None else (lambda n: (B, n, 3))
        x = torch.rand(shp(2500), generator=g).to(cuda).to(dtype).requires_grad_(True)
        y = (torch.rand(shp(2700), generator=g) * 0.7 + 0.2).to(cuda).to(dtype)
        L = SamplesLoss("sinkhorn", p=2, blur=0.05, backend=backend, **kw)
        res = {}
        for one_pass in (True, False):
            monkeypatch.setattr(hip, "_VALUE_GRAD_MIN_PAIRS", 0.0 if one_pass else 1e30)
            ss.set_iteration_fusion(False)        # the fused iteration kernel has its own fused last step
            try:
                v = L(x, y).sum()
                (gx,) = torch.autograd.grad(v, [x])
                res[one_pass] = (v.item(), gx.float())
            finally:
                ss.set_iteration_fusion(True)
        (v1, g1), (v0, g0) = res[True], res[False]
        assert abs(v1 - v0) <= 3e-6 * abs(v0) + 1e-9, (backend, kw, v1, v0)
        gtol = 3e-5 if dtype == torch.float32 else 2 ** -7
        assert (g1 - g0).abs().max().item() <= gtol * g0.abs().max().item() + 1e-9
    # the one-pass path really ran
    calls = []
    orig = hip.softmin_fwd_grad_raw
    monkeypatch.setattr(hip, "softmin_fwd_grad_raw", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    monkeypatch.setattr(hip, "_VALUE_GRAD_MIN_PAIRS", 0.0)
    ss.set_iteration_fusion(False)
    try:
        x = torch.rand(900, 3, generator=g).to(cuda).requires_grad_(True)
        SamplesLoss("sinkhorn", p=2, blur=0.05, backend=backend, **kw)(x, torch.rand(800, 3, generator=g).to(cuda)).backward()
    finally:
        ss.set_iteration_fusion(True)
    assert len(calls) == (2 if kw.get("debias", True) else 1)
