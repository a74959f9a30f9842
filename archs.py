"""INN architectures of the sin-inn path, MI355X build.  Drop-in for the reference's archs.py: same factory
names and call signature ``arch(c, h, w, opt) -> nn.Module`` whose ``forward(x, rev=False)`` returns one tensor
(reference archs.py:19-71 / lit_wrapper.py:17-19).  The graph is wired through the same node/operator protocol
the reference uses from FrEIA, but every operator is the HIP implementation in ``sin-inn_amd/``.
"""
import torch.nn as nn

import sin_inn_amd as Ff           # InputNode / Node / OutputNode / ReversibleGraphNet
import sin_inn_amd as Fm           # IRevNetDownsampling / GLOWCouplingBlock / PermuteRandom

HIDDEN_CHANNELS = 256


def _conv_subnet(c_in, c_out, k):
    pad = k // 2
    return nn.Sequential(nn.Conv2d(c_in, HIDDEN_CHANNELS, k, padding=pad), nn.ReLU(),
                         nn.Conv2d(HIDDEN_CHANNELS, c_out, k, padding=pad))


def subnet_conv(c_in, c_out):
    """3x3 -> ReLU -> 3x3 subnet with 256 hidden channels (reference archs.py:11-13)."""
    return _conv_subnet(c_in, c_out, 3)


def subnet_conv_1x1(c_in, c_out):
    """1x1 variant (reference archs.py:15-17)."""
    return _conv_subnet(c_in, c_out, 1)


def subnet_fc(c_in, c_out):
    """Unused by the reference graph (archs.py:7-9); kept for API parity.  Not supported by the HIP coupling block."""
    return nn.Sequential(nn.Linear(c_in, 512), nn.ReLU(), nn.Linear(512, c_out))


class UncondSRFlow:
    """Unconditional SR-flow style INN (reference archs.py:19-71).

    squeeze_init, then per level: squeeze, ``num_coupling`` x (GLOW coupling with clamp 1.2 whose subnet alternates
    3x3 / 1x1, followed by a fixed random channel permutation seeded with the block index).
    Calling the class returns the ReversibleGraphNet, exactly like the reference's ``__new__`` trick.
    """
    CLAMP = 1.2

    def __new__(cls, c, h, w, opt):
        chain = [Ff.InputNode(c, h, w, name='input')]

        def add(op, kwargs, name):
            chain.append(Ff.Node(chain[-1], op, kwargs, name=name))

        add(Fm.IRevNetDownsampling, {}, 'squeeze_init')
        levels = (opt.scale - 1).bit_length()
        for level in range(levels):
            add(Fm.IRevNetDownsampling, {}, f'squeeze_{level}')
            for blk in range(opt.num_coupling):
                ctor = subnet_conv if blk % 2 == 0 else subnet_conv_1x1
                add(Fm.GLOWCouplingBlock, {'subnet_constructor': ctor, 'clamp': cls.CLAMP}, f'glow_{level}_{blk}')
                add(Fm.PermuteRandom, {'seed': blk}, f'permute_{level}_{blk}')
        chain.append(Ff.OutputNode(chain[-1], name='output'))
        return Ff.ReversibleGraphNet(chain, verbose=False)


# IRN architecture (reference archs.py:74-233): HIP implementation in sin-inn_amd/irn.py, same class names,
# constructor signatures, parameter names and initialisation as the reference.
from sin_inn_amd.irn import DenseBlock, HaarDownsampling, InvBlockExp, InvRescaleNet   # noqa: E402,F401
