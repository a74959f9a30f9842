"""sin-inn_amd -- MI355X (gfx950) native kernels + host glue for the sin-inn single-video INN training path.

Importable as ``sin_inn_amd`` (see sin_inn_amd.py at the repository root: the directory name carries a
hyphen).  Everything numerical runs in ``libsininn.so`` (hand-written HIP, C ABI in include/sininn.h).
"""
from . import _lib, ops                                                     # noqa: F401
from .modules import GLOWCouplingBlock, IRevNetDownsampling, PermuteRandom   # noqa: F401
from .framework import InputNode, Node, OutputNode, ReversibleGraphNet      # noqa: F401
from .optim import FusedAdam                                                 # noqa: F401
from . import irn                                                             # noqa: F401
from . import functional                                                     # noqa: F401
from . import flowloss                                                       # noqa: F401

__all__ = ['GLOWCouplingBlock', 'IRevNetDownsampling', 'PermuteRandom', 'InputNode', 'Node', 'OutputNode',
           'ReversibleGraphNet', 'FusedAdam', 'functional', 'ops']
