"""Runs the reference's own benchmark protocol on the drop-in (GPU box).

Protocol restated from examples/performances/plot_benchmarks_samplesloss_3D.py:24-111,206-231 of the reference:
D = 3; clouds sampled non-uniformly on a sphere of diameter 1 (randn, offset, projected), |randn| normalised weights;
time `L = Loss(a, x, b, y); L.backward()` after one warm-up call, 100 / 10 / 1 loops, stop a backend at 10 s per call;
configs ("gaussian", blur=.1, truncate=3), ("energy",), ("sinkhorn", p=2, blur=.05, diameter=1),
("sinkhorn", p=2, blur=.01, diameter=1)  x  backends tensorized / online / multiscale.
The reference publishes no numbers for this protocol (outputs are produced at doc-build time); this script
produces ours.  Output: gpurun_out/reference_protocol.txt
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss

dev = torch.device("cuda:0")
D = 3
NS = [100, 1000, 10000, 100000, 1000000] if "--quick" in sys.argv else \
     [100, 200, 500, 1000, 2000, 5000, 10000, 20000, 50000, 100000, 200000, 500000, 1000000]
MAXTIME, REDTIME = 10, 2


def generate_samples(N):
    g = torch.Generator(device="cpu").manual_seed(N)
    x = torch.randn(N, D, generator=g); x[:, 0] += 1; x = x / (2 * x.norm(dim=1, keepdim=True))
    y = torch.randn(N, D, generator=g); y[:, 1] += 2; y = y / (2 * y.norm(dim=1, keepdim=True))
    a = torch.randn(N, generator=g).abs(); b = torch.randn(N, generator=g).abs()
    return (a / a.sum()).to(dev), x.to(dev).requires_grad_(True), (b / b.sum()).to(dev), y.to(dev)


def benchmark(loss, N, loops):
    a, x, b, y = generate_samples(N)
    def run():
        L = loss(a, x, b, y); L.backward(); x.grad = None
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(loops): run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / loops


CONFIGS = [("gaussian", dict(blur=0.1, truncate=3)), ("energy", dict()),
           ("sinkhorn", dict(p=2, blur=0.05, diameter=1)), ("sinkhorn", dict(p=2, blur=0.01, diameter=1))]
lines = [f"# {torch.cuda.get_device_name(0)}; seconds per (loss + backward); '-' = not run (tensorized memory / 10 s rule)",
         f"{'config':44s} {'backend':11s} " + " ".join(f"{n:>9d}" for n in NS)]
FIRST = int(sys.argv[sys.argv.index("--from") + 1]) if "--from" in sys.argv else 0   # resume at this (config, backend) row
os.makedirs("gpurun_out", exist_ok=True)
k = -1
for name, kw in CONFIGS:
    for backend in ("tensorized", "online", "multiscale"):
        k += 1
        if k < FIRST:
            continue
        loss = SamplesLoss(name, backend=backend, **kw)
        row, loops, stop = [], 100, False
        for N in NS:
            if stop or (backend == "tensorized" and N > 20000):
                row.append("        -"); continue
            try:
                t = benchmark(loss, N, loops)
            except (RuntimeError, torch.OutOfMemoryError) as e:
                row.append("      err"); stop = True; continue
            row.append(f"{t:9.5f}")
            if loops * t > REDTIME: loops = max(1, loops // 10)
            if t > MAXTIME: stop = True
        line = f"{name + ' ' + str(kw):44s} {backend:11s} " + " ".join(row)
        print(line, flush=True); lines.append(line)
        open("gpurun_out/reference_protocol.txt", "w").write("\n".join(lines) + "\n")
