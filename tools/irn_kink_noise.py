"""How much gradient error do LeakyReLU kink flips cause?  CPU experiment (float64 oracle, no GPU): the IRN oracle at BASELINE
configs[1]'s shape (256x256, -c 4, lr_window 10, batch 2) is evaluated once exactly and once with every conv output perturbed by
eps x max|output| of Gaussian noise (what an fp32 conv with a different summation order does at eps ~ 1e-7 .. 1e-6); the noise
itself is far below the test tolerance, but units whose pre-activation lies within it change slope (1 <-> 0.2):
    eps 1e-07: dx L2 5.2e-05  per-tensor median 5.2e-05  90% 7.4e-05  max 2.5e-03  flat 2.1e-04
    eps 1e-06: dx L2 1.7e-04  per-tensor median 1.7e-04  90% 2.2e-03  max 4.9e-03  flat 7.2e-04
    eps 3e-06: dx L2 2.7e-04  per-tensor median 2.7e-04  90% 3.7e-03  max 7.8e-03  flat 1.1e-03
The HIP path against the fp32 oracle at the same shape: median 1.0e-04, max 5.3e-03, flat 3.5e-04 -- the eps 1e-7 .. 1e-6 band.
tests/test_gpu_irn.py::test_irn_at_baseline_config_shape_matches_oracle takes its bounds from this table."""
import os, sys, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import archs
from oracle import sininn_oracle as O
torch.set_num_threads(8)
opt = types.SimpleNamespace(scale=4, num_coupling=4, lr_dims=84)
torch.manual_seed(11)
net = archs.InvRescaleNet(3, 256, 256, opt)
g5 = torch.Generator().manual_seed(12)
for m in net.modules():
    if isinstance(m, archs.DenseBlock):
        m.conv5.weight.data = torch.randn(m.conv5.weight.shape, generator=g5) * 0.02
sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
x = torch.rand(2, 3, 256, 256); wgt = torch.randn(2, 192, 32, 32)
def run(eps):
    ref = O.IRNOracle(3, 84, scale=4, num_coupling=4)
    O.load_reference_irn_state(ref, sd)
    ref = ref.double()
    gen = torch.Generator().manual_seed(3)
    hooks = []
    if eps:
        for m in ref.modules():
            if isinstance(m, torch.nn.Conv2d):
                hooks.append(m.register_forward_hook(lambda mod, i, o: o + eps * float(o.abs().max()) * torch.randn(o.shape, generator=gen, dtype=o.dtype)))
    xc = x.double().detach().clone(); xc.requires_grad_(True)
    (ref(xc) * wgt.double()).sum().backward()
    return xc.grad, {n: p.grad for n, p in ref.named_parameters()}
l2 = lambda u, v: float((u - v).norm() / v.norm())
clean = run(0.0)
for eps in (1e-7, 1e-6, 3e-6):
    noisy = run(eps)
    per = sorted(l2(noisy[1][n], clean[1][n]) for n in clean[1])
    fa = torch.cat([noisy[1][n].reshape(-1) for n in clean[1]]); fb = torch.cat([clean[1][n].reshape(-1) for n in clean[1]])
    print(f'eps {eps:g}: dx L2 {l2(noisy[0], clean[0]):.2e}  per-tensor median {per[len(per)//2]:.2e}  90% {per[len(per)*9//10]:.2e}  max {per[-1]:.2e}  flat {l2(fa, fb):.2e}', flush=True)
