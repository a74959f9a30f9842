"""Generate the golden fixtures in this directory FROM THE REFERENCE'S OWN CODE.

Run once in the build container (needs /root/reference; never runs on the GPU box):
    python tests/golden/make_golden.py
It imports the reference's loss.py unmodified and archs.py with empty namespace modules registered
for the absent third-party package FrEIA (archs.py:4-5 only needs the names to exist at import time;
nothing of FrEIA is emulated -- UncondSRFlow is NOT exercised).  Outputs are data only
(inputs, weights, expected outputs) written to golden_*.npz.
"""
import os, sys, types
import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    for name in ('FrEIA', 'FrEIA.framework', 'FrEIA.modules'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    import archs as ref_archs, loss as ref_loss          # noqa: E401
    sys.path.pop(0)
    return ref_archs, ref_loss


def t2n(d):
    return {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def main():
    archs, loss = import_reference()
    out = {}

    # G1 losses (loss.py:3-5, 38-39)
    g = torch.Generator().manual_seed(11)
    x, y = torch.randn(2, 5, 6, 7, generator=g), torch.randn(2, 5, 6, 7, generator=g)
    out.update(g1_x=x, g1_y=y, g1_rec=loss.reconstruction(x, y), g1_nll=loss.latent_nll(x))

    # G2 HaarDownsampling (archs.py:162-199)
    haar = archs.HaarDownsampling(3)
    x = torch.randn(2, 3, 16, 16, generator=g)
    yh = haar(x)
    out.update(g2_x=x, g2_fwd=yh, g2_jac_fwd=np.float64(haar.last_jac))
    xr = haar(yh, rev=True)
    out.update(g2_rev=xr, g2_jac_rev=np.float64(haar.last_jac))

    # G3 conv subnets (archs.py:11-17)
    torch.manual_seed(3)
    for tag, ctor in (('3x3', archs.subnet_conv), ('1x1', archs.subnet_conv_1x1)):
        net = ctor(24, 48)
        x = torch.randn(2, 24, 16, 16, generator=g)
        out.update({f'g3_{tag}_x': x, f'g3_{tag}_y': net(x),
                    f'g3_{tag}_w0': net[0].weight, f'g3_{tag}_b0': net[0].bias,
                    f'g3_{tag}_w2': net[2].weight, f'g3_{tag}_b2': net[2].bias})

    # G4 DenseBlock / InvBlockExp with a re-seeded NON-ZERO conv5 (default init is the identity)
    torch.manual_seed(4)
    blk = archs.InvBlockExp(8, 4)
    for db in (blk.F, blk.G, blk.H):
        torch.nn.init.normal_(db.conv5.weight, std=0.05)
        torch.nn.init.normal_(db.conv5.bias, std=0.05)
    x = torch.randn(2, 8, 8, 8, generator=g)
    yb = blk(x)
    out.update(g4_x=x, g4_fwd=yb, g4_rev=blk(yb, rev=True), g4_dense_F=blk.F(x[:, 4:]))
    for k, v in blk.state_dict().items():
        out['g4_sd_' + k] = v

    # G5 InvRescaleNet at config-1 shape: weights regenerated from the seed at test time,
    # conv5 of every DenseBlock re-drawn from generator 55 (so the net is not the identity)
    opt = types.SimpleNamespace(scale=4, num_coupling=4, lr_dims=12)
    torch.manual_seed(5)
    net = archs.InvRescaleNet(3, 64, 64, opt)
    g5 = torch.Generator().manual_seed(55)
    for m in net.modules():
        if isinstance(m, archs.DenseBlock):
            m.conv5.weight.data = torch.randn(m.conv5.weight.shape, generator=g5) * 0.02
    x = torch.rand(2, 3, 64, 64, generator=g)
    with torch.no_grad():
        yn = net(x)
        xr = net(yn, rev=True)
    out.update(g5_x=x, g5_out_slice=yn[:, ::16, ::2, ::2], g5_out_norm=yn.norm(), g5_out_sum=yn.sum(),
               g5_roundtrip_err=(xr - x).abs().max(),
               g5_nparams=np.int64(sum(p.numel() for p in net.parameters())))

    # G6 legacy numpy permutation stream (what FrEIA's PermuteRandom draws: archs.py:65-68)
    for c in (48, 192):
        for k in range(12):
            np.random.seed(k)
            out[f'g6_perm_{c}_{k}'] = np.random.permutation(c)

    # G7 loss.mmd (loss.py:9-36), forward and reverse kernel sets, values + gradients.  The function hard-codes
    # `.to('cuda')` for three zero buffers (loss.py:27-29); for the duration of the call Tensor.to is
    # wrapped so that a 'cuda' target is a no-op -- the arithmetic that runs is the reference's own, on CPU.
    real_to = torch.Tensor.to

    def cpu_to(self, *a, **k):
        if a and isinstance(a[0], str) and a[0].startswith('cuda'):
            return self
        return real_to(self, *a, **k)

    torch.Tensor.to = cpu_to
    try:
        for tag, shape, spread in (('a', (4, 3, 5, 6), 0.35), ('b', (6, 8, 4, 4), 0.15)):
            x = (torch.randn(*shape, generator=g) * spread).requires_grad_(True)
            y = (torch.randn(*shape, generator=g) * spread).requires_grad_(True)
            out.update({f'g7_{tag}_x': x.detach().clone(), f'g7_{tag}_y': y.detach().clone()})
            for rev in (False, True):
                x.grad = y.grad = None
                val = loss.mmd(x, y, rev=rev)
                val.backward()
                r = 'rev' if rev else 'fwd'
                out.update({f'g7_{tag}_{r}': val.detach().clone(), f'g7_{tag}_{r}_gx': x.grad.clone(),
                            f'g7_{tag}_{r}_gy': y.grad.clone()})
    finally:
        torch.Tensor.to = real_to

    np.savez_compressed(os.path.join(HERE, 'golden_reference.npz'), **t2n(out))
    print('wrote', os.path.join(HERE, 'golden_reference.npz'),
          os.path.getsize(os.path.join(HERE, 'golden_reference.npz')), 'bytes')


if __name__ == '__main__':
    main()
