"""Prints the relative error of the HIP backends against the reference's golden outputs (GPU box), per golden case and per
kernel-selection flag; exits non-zero on any exception (tools/final_measure.sh redirects stdout to gpurun_out/accuracy_report.txt
and refuses to keep a traceback as a report)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import golden_cases, load_golden, relerr
from geomloss_amd import SamplesLoss, hip

dev = torch.device("cuda:0")
lines = []


def run(rec, backend, flags):
    hip.ENV_FLAGS = flags
    a, x, b, y = (torch.from_numpy(rec[k]).float().to(dev) for k in "axby")
    x.requires_grad_(True)
    L = SamplesLoss(backend=backend, **rec["kwargs"])(a, x, b, y)
    (gx,) = torch.autograd.grad(L.sum(), [x])
    F, G = SamplesLoss(backend=backend, potentials=True, **rec["kwargs"])(a, x.detach(), b, y)
    return L.detach().cpu().numpy(), gx.cpu().numpy(), F.cpu().numpy()


hdr = f"{'case':24s} {'variant':14s} {'loss vs ref f64':>16s} {'loss vs ref f32':>16s} {'grad_x vs f64':>14s} {'potential vs f64':>17s}"
lines.append(hdr)
for name in golden_cases():
    rec = load_golden(name)
    for label, flags in (("default", 0), ("no-mfma", 2), ("direct", 3)):
        L, gx, F = run(rec, "online", flags)
        lines.append(f"{name:24s} {label:14s} {relerr(L, rec['loss_f64']):16.2e} {relerr(L, rec['loss_f32']):16.2e} "
                     f"{relerr(gx, rec['gx_f64']):14.2e} {relerr(F, rec['F_f64']):17.2e}")
    lines.append(f"{name:24s} {'(ref f32 run)':14s} {relerr(rec['loss_f32'], rec['loss_f64']):16.2e} {'':16s} "
                 f"{relerr(rec['gx_f32'], rec['gx_f64']):14.2e} {relerr(rec['F_f32'], rec['F_f64']):17.2e}")
hip.ENV_FLAGS = 0
rec = load_golden("cfg1_n2000_d2")
x, y = torch.from_numpy(rec["x"]).to(dev), torch.from_numpy(rec["y"]).to(dev)
for label, flags in (("default", 0), ("no-mfma", 2), ("direct", 3)):
    hip.ENV_FLAGS = flags
    L = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="online")(x, y).item()
    lines.append(f"cfg1 N=M=2000 2D same-law  {label:10s} loss {L:.8e}  vs ref f64 {abs(L-float(rec['loss_f64']))/float(rec['loss_f64']):.2e}"
                 f"  (ref f32 vs f64 {abs(float(rec['loss_f32'])-float(rec['loss_f64']))/float(rec['loss_f64']):.2e})")
hip.ENV_FLAGS = 0
out = "\n".join(lines)
print(out)
