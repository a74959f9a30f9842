"""Graph front end with the FrEIA-0.1 surface the reference uses (archs.py:26-71):
``InputNode(c,h,w,name=)``, ``Node(prev, ModuleClass, kwargs, name=)``, ``OutputNode(prev, name=)``,
``ReversibleGraphNet(node_list, verbose=False)``; ``net(x, rev=False)`` returns ONE tensor
(lit_wrapper.py:45-46 slices it directly).

MI355X-first execution: the node list of the sin-inn path is a straight chain, so instead of
interpreting it op by op the net is lowered once into a short list of fused launches:
  * consecutive IRevNetDownsampling nodes (and the NCHW -> pixel-major import) become one gather pass,
  * every PermuteRandom is folded into the store addressing of the GLOW coupling kernel that
    produces its input (forward: dst = perm_inv; reverse: the producer stores through perm),
so squeezes/permutes cost no extra HBM passes.  Parameters keep the FrEIA state-dict layout
``module_list.<node index>.s1.0.weight`` so reference checkpoints load.
"""
import torch
import torch.nn as nn

from .modules import (GLOWCouplingBlock, IRevNetDownsampling, PermuteRandom, import_nchw, squeeze_op)


class _NoOp(nn.Module):
    """Placeholder module of Input/Output nodes (keeps module_list indices == node indices)."""

    def forward(self, x, rev=False):
        return x


class Node:
    def __init__(self, inputs, module_type, module_args, conditions=[], name=None):
        if not isinstance(inputs, (list, tuple)):
            inputs = [inputs]
        self.inputs = [i[0] if isinstance(i, tuple) else i for i in inputs]
        self.module_type, self.module_args, self.name = module_type, module_args, name
        assert not conditions, 'conditional nodes are not on the sin-inn path'
        self.input_dims = None
        self.module = None
        self.output_dims = None
        self.out0 = (self, 0)

    def build(self):
        if self.module is None:
            self.input_dims = [n.output_dims[0] for n in self.inputs]
            self.module = self.module_type(self.input_dims, **self.module_args)
            self.output_dims = self.module.output_dims(self.input_dims)
        return self.module


class InputNode(Node):
    def __init__(self, *dims, name='node'):
        self.inputs, self.name = [], name
        self.module = _NoOp()
        self.output_dims = [tuple(dims)]
        self.input_dims = []
        self.out0 = (self, 0)

    def build(self):
        return self.module


class OutputNode(Node):
    def __init__(self, inp, name='node'):
        self.inputs = [inp[0] if isinstance(inp, tuple) else inp]
        self.name = name
        self.module = _NoOp()
        self.output_dims = None
        self.out0 = (self, 0)

    def build(self):
        self.output_dims = self.inputs[0].output_dims
        return self.module


class ReversibleGraphNet(nn.Module):
    def __init__(self, node_list, ind_in=None, ind_out=None, verbose=False):
        super().__init__()
        self.node_list = node_list
        ins = [n for n in node_list if isinstance(n, InputNode)]
        outs = [n for n in node_list if isinstance(n, OutputNode)]
        assert len(ins) == 1 and len(outs) == 1, 'exactly one input and one output node expected'
        # the sin-inn graphs are chains: verify and order input -> output
        order, cur = [outs[0]], outs[0]
        while cur.inputs:
            assert len(cur.inputs) == 1, 'branching graphs are not on the sin-inn path'
            cur = cur.inputs[0]
            order.append(cur)
        order.reverse()
        assert order[0] is ins[0] and len(order) == len(node_list), 'graph is not a single chain'
        self.chain = order
        self.module_list = nn.ModuleList([n.build() for n in node_list])
        self._plans = {}
        if verbose:
            for n in order:
                print(n.name, n.output_dims)

    # ---- lowering --------------------------------------------------------------------------------
    def _lower(self, rev):
        """Chain of modules -> list of fused steps (kind, payload)."""
        mods = [n.module for n in self.chain if not isinstance(n.module, _NoOp)]
        for m in mods:
            if not isinstance(m, (GLOWCouplingBlock, PermuteRandom, IRevNetDownsampling)):
                raise NotImplementedError(f'{type(m).__name__} has no HIP lowering in sin-inn_amd')
        steps = []
        if not rev:
            i = 0
            while i < len(mods):
                m = mods[i]
                if isinstance(m, IRevNetDownsampling):
                    lv = 0
                    while i < len(mods) and isinstance(mods[i], IRevNetDownsampling):
                        lv += 1; i += 1
                    steps.append(('squeeze', dict(levels=lv, inverse=False, perm=None)))
                elif isinstance(m, GLOWCouplingBlock):
                    perm = None
                    if i + 1 < len(mods) and isinstance(mods[i + 1], PermuteRandom):
                        perm = (mods[i + 1], 'inv')          # store channel c at perm_inv[c]
                        i += 1
                    steps.append(('glow', dict(block=m, perm=perm)))
                    i += 1
                else:
                    steps.append(('permute', dict(mod=m)))
                    i += 1
        else:
            seq = mods[::-1]
            i = 0
            while i < len(seq):
                m = seq[i]
                if isinstance(m, PermuteRandom):
                    # z[:, c] = y[:, perm_inv[c]]: fold into the producer of y when there is one
                    if steps and steps[-1][0] in ('glow', 'squeeze') and steps[-1][1]['perm'] is None:
                        steps[-1][1]['perm'] = (m, 'fwd')    # producer stores channel j at perm[j]
                    else:
                        steps.append(('permute', dict(mod=m)))
                    i += 1
                elif isinstance(m, IRevNetDownsampling):
                    lv = 0
                    while i < len(seq) and isinstance(seq[i], IRevNetDownsampling):
                        lv += 1; i += 1
                    steps.append(('squeeze', dict(levels=lv, inverse=True, perm=None)))
                else:
                    steps.append(('glow', dict(block=m, perm=None)))
                    i += 1
        return steps

    def forward(self, x, c=None, rev=False):
        if isinstance(x, (list, tuple)):
            x = x[0]
        if not x.is_cuda:
            raise NotImplementedError('sin-inn_amd runs on the GPU only: move the module and its inputs to cuda')
        if rev not in self._plans:
            self._plans[rev] = self._lower(rev)
        steps = self._plans[rev]
        dev = x.device
        cur = x                                   # (B,C,H,W)-shaped, any strides
        pixel_major = False                       # True once `cur` is a contiguous (B,H,W,C) tensor
        n_glow = sum(1 for kind, _ in steps if kind == 'glow')
        ld_rows = iter(torch.zeros(n_glow, x.shape[0], device=dev, dtype=torch.float32)) if n_glow else None
        for kind, p in steps:
            cmap = None
            if p.get('perm') is not None:
                mod, which = p['perm']
                perm, perm_inv = mod.maps(dev)
                cmap = perm_inv if which == 'inv' else perm
            if kind == 'squeeze':
                src = cur.permute(0, 3, 1, 2) if pixel_major else cur
                cur = squeeze_op(src, p['levels'], inverse=p['inverse'], chan_map=cmap,
                                 map_on_out=cmap is not None).permute(0, 2, 3, 1)
                pixel_major = True
            elif kind == 'glow':
                if not pixel_major:
                    cur = import_nchw(cur)
                    pixel_major = True
                p['block']._ld_row = next(ld_rows)
                cur = p['block'].apply_pixel_major(cur, rev=rev, dst=cmap)
            else:
                src = cur.permute(0, 3, 1, 2) if pixel_major else cur
                perm, perm_inv = p['mod'].maps(dev)
                cur = squeeze_op(src, 0, inverse=False, chan_map=perm_inv if rev else perm,
                                 map_on_out=False).permute(0, 2, 3, 1)
                pixel_major = True
        return cur.permute(0, 3, 1, 2) if pixel_major else cur

    def set_precision(self, precision):
        """'fp32' (default, the reference's arithmetic) or 'bf16' (mixed precision: bf16 conv subnets, fp32 flow)."""
        assert precision in ('fp32', 'bf16')
        for m in self.module_list:
            if isinstance(m, GLOWCouplingBlock):
                m.precision = precision
        return self

    def prepare_packs(self):
        """Build (or refresh) every packed-weight buffer of the graph on the CURRENT stream.  Packs are otherwise built
        lazily by the first pass that misses the cache, on that pass's stream; with the forward and the reverse chain on
        two streams the other chain would then read half-written packs.  lit_wrapper.training_step calls this on the main
        stream before it forks the second one (cache hits cost nothing)."""
        from .modules import _subnet_args
        for m in self.module_list:
            if isinstance(m, GLOWCouplingBlock):
                dev = m.s1[0].weight.device
                _subnet_args(m, m.s1, m.split_len2, dev, True, False)
                _subnet_args(m, m.s2, m.split_len1, dev, True, False)

    @property
    def concurrent_passes_safe(self):
        """True when two passes (forward + backward each) may run on two streams at once: every parameter gradient of
        this graph is accumulated by the GLOW block executor on the dedicated weight-gradient stream."""
        from .modules import USE_SIDE_STREAM
        return bool(USE_SIDE_STREAM[0])

    def log_jacobian(self, x=None, c=None, rev=False, run_forward=True):
        if run_forward and x is not None:
            self.forward(x, rev=rev)
        tot = 0.
        for m in self.module_list:
            if isinstance(m, GLOWCouplingBlock) and m.last_jac is not None:
                tot = tot + m.last_jac
        return tot
