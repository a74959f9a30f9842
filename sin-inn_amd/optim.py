"""Fused Adam for the INN parameters (replaces torch.optim.Adam of lit_wrapper.py:131-138).

All parameters are re-homed into ONE flat fp32 buffer (and their gradients into another) so that
  * the optimiser step is a single HIP launch over 3.7 M elements (p, g, m, v streams), and
  * data-parallel training needs a single RCCL all-reduce of the flat gradient per step.
Arithmetic == torch.optim.Adam (L2 weight decay added to the gradient, bias corrections, eps outside
the sqrt); parity is tested against torch.optim.Adam itself.
"""
import torch

from . import ops
from .modules import bump_weights_epoch, join_side_streams


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._flat = []
        for group in self.param_groups:
            ps = [p for p in group['params'] if p.requires_grad]
            assert ps, 'empty parameter group'
            dev = ps[0].device
            if dev.type != 'cuda':
                raise NotImplementedError('FusedAdam runs on the GPU only: move the module to cuda first')
            n = sum(p.numel() for p in ps)
            npad = (n + 3) // 4 * 4
            flat_p = torch.zeros(npad, device=dev, dtype=torch.float32)
            flat_g = torch.zeros(npad, device=dev, dtype=torch.float32)
            off = 0
            for p in ps:
                assert p.dtype == torch.float32 and p.device == dev
                k = p.numel()
                flat_p[off:off + k].copy_(p.detach().reshape(-1))
                had_grad = p.grad
                p.data = flat_p[off:off + k].view(p.shape)             # parameter storage -> flat buffer
                p.grad = flat_g[off:off + k].view(p.shape)             # gradient storage  -> flat buffer
                if had_grad is not None:
                    p.grad.copy_(had_grad)
                off += k
            self._flat.append(dict(p=flat_p, g=flat_g, m=torch.zeros_like(flat_p), v=torch.zeros_like(flat_p),
                                   n=n, step=0, params=ps))
        bump_weights_epoch()

    # the flat gradient buffer must survive zero_grad (views would be lost with set_to_none=True)
    def zero_grad(self, set_to_none=False):
        join_side_streams()
        for fl in self._flat:
            fl['g'].zero_()

    def flat_grads(self):
        join_side_streams()                 # weight-gradient kernels run on a side stream
        return [fl['g'] for fl in self._flat]

    def flat_grads_nojoin(self):
        """the flat gradient buffers, no stream join (inside a graph capture the caller has joined already)"""
        return [fl['g'] for fl in self._flat]

    def flat_grad_buffers(self):
        """The flat gradient buffers and the stream a collective over them has to be ordered behind.  That is the weight-gradient
        stream ALONE (no join; dist.allreduce_sum_(..., after=stream)) only while EVERY parameter of the buffers is an
        executor-owned conv parameter, i.e. all writes into the buffers are issued on that stream (modules._grad_buf tags them).
        A model with any other trainable parameter -- a torch-native layer whose gradient autograd writes on the main or the
        second pass stream -- gets (buffers, None) after a join of the streams: the collective is then ordered behind the
        caller's stream, which has waited for everything."""
        from .modules import side_stream_if_any
        bufs = [fl['g'] for fl in self._flat]
        if not self.__dict__.get('_all_executor_owned', False):
            self.__dict__['_all_executor_owned'] = all(p.__dict__.get('_sininn_executor_grad', False)
                                                       for fl in self._flat for p in fl['params'])
        if not self._all_executor_owned:
            join_side_streams()
            return bufs, None
        return bufs, side_stream_if_any(bufs[0].device)

    def flat_params(self):
        return [fl['p'] for fl in self._flat]

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        join_side_streams()
        for group, fl in zip(self.param_groups, self._flat):
            lo, hi = fl['g'].data_ptr(), fl['g'].data_ptr() + fl['g'].numel() * 4
            for p in fl['params']:
                if p.grad is None or not (lo <= p.grad.data_ptr() < hi):
                    raise RuntimeError('FusedAdam: a parameter gradient left the flat buffer '
                                       '(use this optimizer\'s zero_grad(), not set_to_none)')
            fl['step'] += 1
            b1, b2 = group['betas']
            ops.adam_step(fl['p'], fl['g'], fl['m'], fl['v'], group['lr'], b1, b2, group['eps'],
                          group['weight_decay'], fl['step'], grad_scale)
        bump_weights_epoch()
        return loss

    def state_dict(self):
        return {'flat': [dict(m=fl['m'].clone(), v=fl['v'].clone(), step=fl['step']) for fl in self._flat],
                'param_groups': [{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups]}

    def load_state_dict(self, sd):
        for fl, s in zip(self._flat, sd['flat']):
            fl['m'].copy_(s['m']); fl['v'].copy_(s['v']); fl['step'] = int(s['step'])
        for g, s in zip(self.param_groups, sd['param_groups']):
            g.update(s)
