"""Diagnostic: per-tensor max-norm gradient error of the cfg-1-shape training step vs the CPU oracle, for several
weight-gradient paths (sininn_wgrad_test_hooks): 0 grouped Winograd, 16 per-conv Winograd, 2 direct (no Winograd)."""
import os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import lit_wrapper
from data import FrameStore
from oracle import sininn_oracle as O
from sin_inn_amd import _lib
from sin_inn_amd.functional import sample_windows
from test_gpu_model import make_opt

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.set_num_threads(16)
opt = make_opt(num_coupling=4, lr_window=10)
store = FrameStore.synthetic(40, 256, 256)
g = torch.Generator().manual_seed(6)
idx = torch.randint(10, 30, (batch,), generator=g)
pairs = [O.gather_window(store.lr, store.hr, i, 10) for i in idx.tolist()]
hr_c, lr_c = torch.stack([p[0] for p in pairs]), torch.stack([p[1] for p in pairs])
z = torch.randn(batch, opt.z_dims, 32, 32, generator=g)
lam = dict(fwd_rec=1.0, fwd_mmd=0.0, latent_nll=0.0, bwd_rec=1.0, bwd_mmd=0.0)
torch.manual_seed(21)
ref = O.SRFlowOracle(3, 256, 256, scale=4, num_coupling=4)
O.training_step(ref, hr_c, lr_c, z, lam, opt.lr_dims)
# float64 oracle as the arbiter of which fp32 result is closer to the truth
ref64 = O.SRFlowOracle(3, 256, 256, scale=4, num_coupling=4).double()
ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
O.training_step(ref64, hr_c.double(), lr_c.double(), z.double(), lam, opt.lr_dims)
names = [n for n, _ in ref.named_parameters()]
g32 = [p.grad.reshape(-1) for p in ref.parameters()]
g64 = [p.grad.reshape(-1) for p in ref64.parameters()]
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
print('torch-CPU fp32 oracle vs float64: worst per-tensor', max(rel(a, b) for a, b in zip(g32, g64)))
lit_wrapper._latent = lambda b, zd, h, w, device, temp=1.0: z.to(device)
for hook in (0, 16, 2):
    _lib.lib().sininn_wgrad_test_hooks(hook)
    model = lit_wrapper.SingleVideoINN(3, 256, 256, opt)
    model.load_state_dict({'inn.' + k: v for k, v in ref.state_dict().items()})
    model.cuda()
    optim = model.attach_optimizer()
    hr_g, lr_g = sample_windows(store.hr.cuda(), store.lr.cuda(), idx.cuda(), 10)
    model.training_step([{'hr': hr_g, 'lr': lr_g}, {'hr': hr_g, 'lr': lr_g}], 0)
    flat = optim.flat_grads()[0].cpu()
    off, worst32, worst64 = 0, (0, ''), (0, '')
    for n, a, b in zip(names, g32, g64):
        k = a.numel()
        e32, e64 = rel(flat[off:off + k], a), rel(flat[off:off + k], b)
        worst32, worst64 = max(worst32, (e32, n)), max(worst64, (e64, n))
        off += k
    print(f'hook {hook}: worst per-tensor err vs fp32 oracle {worst32[0]:.2e} ({worst32[1]}), vs float64 {worst64[0]:.2e} ({worst64[1]}); '
          f'whole flat vs fp32 {rel(flat[:off], torch.cat(g32)):.2e} vs f64 {rel(flat[:off], torch.cat(g64)):.2e}')
_lib.lib().sininn_wgrad_test_hooks(0)
