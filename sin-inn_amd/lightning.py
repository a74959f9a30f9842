"""Minimal stand-in for the slice of pytorch_lightning 1.2 that the reference's training loop uses
(main.py:108-118, lit_wrapper.py:12-138, data.py:125-137).  pytorch_lightning is not installed in this image and
its 1.2 API (``Trainer(gpus=...)``, ``ModelCheckpoint(period=...)``) no longer exists upstream, so the surface
is provided here: ``LightningModule`` (optimizers / manual_backward / log / save_hyperparameters),
``LightningDataModule``, ``Trainer.fit`` (epoch loop, validation every n epochs, checkpoint every ``period``
epochs, resume), ``ModelCheckpoint`` and a file logger in place of WandbLogger.

Data parallel: under ``torchrun`` every rank runs the same loop on its shard of each batch; the optimizer proxy
averages the flat gradient buffer with ONE all-reduce (RCCL) right before ``step()``.
"""
import inspect
import json
import os
import time

import torch
import torch.nn as nn

from . import dist as sdist


class LightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.automatic_optimization = True
        self.trainer = None
        self.hparams = {}
        self._logged = {}

    def save_hyperparameters(self, *_):
        frame = inspect.currentframe().f_back
        args = inspect.getargvalues(frame)
        self.hparams = {k: args.locals[k] for k in args.args if k != 'self'}

    def optimizers(self):
        assert self.trainer is not None, 'optimizers() needs a Trainer (or attach_optimizer())'
        return self.trainer.optimizer

    def manual_backward(self, loss, *a, **kw):
        loss.backward(*a, **kw)

    def log(self, name, value, **_):
        self._logged[name] = value.detach() if torch.is_tensor(value) else value

    @property
    def device(self):
        return next(self.parameters()).device

    def attach_optimizer(self, optimizer=None):
        """Use the module without Trainer.fit (bench.py, tests): builds the optimizer proxy."""
        tr = Trainer.__new__(Trainer)
        tr.optimizer = _OptimizerProxy(optimizer if optimizer is not None else self.configure_optimizers())
        tr.current_epoch, tr.global_step = 0, 0
        self.trainer = tr
        return tr.optimizer


class LightningDataModule:
    def train_dataloader(self):
        raise NotImplementedError

    def val_dataloader(self):
        return None


class _OptimizerProxy:
    """What ``self.optimizers()`` returns: zero_grad()/step() of the wrapped optimizer + the DP gradient all-reduce."""

    def __init__(self, optimizer):
        self.optimizer = optimizer

    def zero_grad(self):
        self.optimizer.zero_grad()

    def step(self, *a, **kw):
        _, ws = sdist.world()
        if ws > 1:
            if hasattr(self.optimizer, 'flat_grad_buffers'):
                # ONE collective on the flat buffer; the mean's 1/world rides on the Adam kernel's gradient scale.
                # (No bucketing: the forward chain's backward walks the blocks last -> first and the reverse chain's
                # first -> last, both accumulating into the same buffer, so no block's gradient is final before the tail
                # of the step -- there is no window to overlap an earlier bucket with.)
                # Ordered behind the weight-gradient stream alone (its last kernel is already queued: both chains have been
                # issued), on RCCL's high-priority stream; this stream -- and with it the Adam launch -- waits for the result.
                bufs, wstream = self.optimizer.flat_grad_buffers()
                sdist.allreduce_sum_(bufs, after=wstream)
                kw = dict(kw, grad_scale=kw.get('grad_scale', 1.0) / ws)
            elif hasattr(self.optimizer, 'flat_grads'):
                sdist.allreduce_sum_(self.optimizer.flat_grads())
                kw = dict(kw, grad_scale=kw.get('grad_scale', 1.0) / ws)
            else:
                sdist.allreduce_mean_([p.grad for g in self.optimizer.param_groups for p in g['params']
                                       if p.grad is not None])
        return self.optimizer.step(*a, **kw)

    def __getattr__(self, name):
        return getattr(self.optimizer, name)


class ModelCheckpoint:
    def __init__(self, period=1, dirpath=None):
        self.period, self.dirpath = max(int(period), 1), dirpath


class FileLogger:
    """JSON-lines logger (stands in for WandbLogger(project='sin-inn', ...), main.py:105-107)."""

    def __init__(self, project='sin-inn', save_dir='.', name='run', **_):
        self.path = os.path.join(save_dir, f'{project}_{name}.jsonl')

    def log_hyperparams(self, args):
        self._write({'hyperparams': {k: (v if isinstance(v, (int, float, str, bool, type(None))) else str(v))
                                     for k, v in vars(args).items()}})

    def log_metrics(self, metrics, step):
        self._write({'step': step, **metrics})

    def _write(self, obj):
        rank, _ = sdist.world()
        if rank == 0:
            with open(self.path, 'a') as f:
                f.write(json.dumps(obj) + '\n')


WandbLogger = FileLogger


def _to_device(batch, device):
    if torch.is_tensor(batch):
        return batch.to(device, non_blocking=True)
    if isinstance(batch, dict):
        return {k: _to_device(v, device) for k, v in batch.items()}
    if isinstance(batch, (list, tuple)):
        return type(batch)(_to_device(v, device) for v in batch)
    return batch


_DROP = object()


def _plain(v):
    """Hyper-parameters as they go into a checkpoint: numbers / strings / None and lists / dicts of those.  Anything else
    (the decoded clip ``opt.frame_store``, modules, tensors) is dropped: a checkpoint holds tensors and primitives only,
    so it loads under torch.load's default ``weights_only=True`` and does not carry the video."""
    if isinstance(v, (bool, int, float, str, type(None))):
        return v
    if isinstance(v, (list, tuple)):
        items = [_plain(x) for x in v]
        return _DROP if any(x is _DROP for x in items) else items
    if isinstance(v, dict):
        return {str(k): x for k, x in ((k, _plain(x)) for k, x in v.items()) if x is not _DROP}
    return _DROP


class _Foreign:
    """Inert stand-in for a global that a checkpoint names and the restricted loader does not know (a Lightning callback class,
    a logger, ...): constructing it, calling it or restoring state into it does nothing."""
    _origin = '?'

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __setstate__(self, state):
        pass


def _is_foreign(v):
    return isinstance(v, _Foreign) or (isinstance(v, type) and issubclass(v, _Foreign))


def _strip_foreign(v):
    """Drop dict entries / list items that are (or are keyed by) foreign stand-ins, recursively."""
    if isinstance(v, dict):
        return type(v)((k, _strip_foreign(x)) for k, x in v.items() if not _is_foreign(k) and not _is_foreign(x))
    if isinstance(v, (list, tuple)):
        return type(v)(_strip_foreign(x) for x in v if not _is_foreign(x))
    return v


class _StubbingPickle:
    """``pickle_module`` for torch.load: tensors, containers, primitives and ``argparse.Namespace`` load as usual; EVERY other
    global resolves to an inert ``_Foreign`` subclass instead of being imported, so nothing from the file can run and a
    checkpoint written by the reference stack (pytorch_lightning 1.2 ``ModelCheckpoint`` stores ``callbacks: {<class>: state}``,
    i.e. pickled class globals as dict keys) loads without pytorch_lightning being importable."""
    __name__ = 'sin_inn_amd.lightning._StubbingPickle'
    _ALLOWED = {('collections', 'OrderedDict'), ('argparse', 'Namespace'), ('torch', 'Size'), ('torch', 'device'),
                ('torch._utils', '_rebuild_tensor'), ('torch._utils', '_rebuild_tensor_v2'),
                ('torch._utils', '_rebuild_parameter'), ('torch._utils', '_rebuild_parameter_with_state'),
                ('torch._tensor', '_rebuild_from_type_v2'), ('torch.nn.parameter', 'Parameter'), ('torch', 'Tensor'),
                ('torch.storage', 'UntypedStorage'), ('torch.storage', 'TypedStorage'), ('torch.storage', '_load_from_bytes'),
                ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'), ('numpy', 'dtype')}
    foreign_seen = None

    import pickle as _pickle

    class Unpickler(_pickle.Unpickler):
        def find_class(self, module, name):
            if (module, name) in _StubbingPickle._ALLOWED:
                return super().find_class(module, name)
            if module == 'torch' and (name.endswith('Storage') or isinstance(getattr(torch, name, None), torch.dtype)):
                return getattr(torch, name)
            if _StubbingPickle.foreign_seen is not None:
                _StubbingPickle.foreign_seen.add(f'{module}.{name}')
            return type(name, (_Foreign,), {'_origin': f'{module}.{name}'})

    @staticmethod
    def load(f, **kw):
        return _StubbingPickle.Unpickler(f, **kw).load()


def load_checkpoint(path, map_location=None, trust=False):
    """torch.load for checkpoints of this trainer (tensors + primitives) and for Lightning checkpoints written by the
    reference stack (main.py:127; pytorch_lightning 1.2: an ``argparse.Namespace`` under 'hyper_parameters' and
    ``callbacks: {<class ModelCheckpoint>: state}``, whose keys are pickled class globals).  First torch's restricted
    unpickler (``weights_only=True``) with ``argparse.Namespace`` allow-listed; a file it refuses is re-read with a loader that
    resolves every unknown global to an inert stand-in (nothing from the file is imported or run) and drops the entries that
    contained one -- so a genuine reference checkpoint resumes without pytorch_lightning installed.  ``trust=True`` (CLI
    ``--trust_checkpoint``) is the explicit opt-in to the full unpickler -- which can run arbitrary code from the file."""
    import argparse
    import logging
    import pickle
    if trust:
        logging.warning(f'{path}: loading with the FULL unpickler (--trust_checkpoint): code inside the file can run')
        return torch.load(path, map_location=map_location, weights_only=False)
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            return torch.load(path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as first:
        _StubbingPickle.foreign_seen = set()
        try:
            ck = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_StubbingPickle)
        except Exception as second:
            raise pickle.UnpicklingError(
                f'{path}: not loadable with the restricted unpickler ({first}) nor with foreign classes stubbed out ({second}). '
                'If you trust the file, pass --trust_checkpoint (full unpickler: code inside the file can run; the packages it '
                'names, e.g. pytorch_lightning, must be importable).') from second
        finally:
            seen, _StubbingPickle.foreign_seen = sorted(_StubbingPickle.foreign_seen), None
        logging.warning(f'{path}: entries referring to classes that are not loaded here were dropped: ' + ', '.join(seen[:6]))
        ck = _strip_foreign(ck)
        if not isinstance(ck, dict) or 'state_dict' not in ck:
            raise pickle.UnpicklingError(f'{path}: no state_dict left after dropping foreign entries; pass --trust_checkpoint '
                                         'if you trust the file (and make the packages it names importable)')
        return ck


class Trainer:
    def __init__(self, gpus=None, max_epochs=1000, check_val_every_n_epoch=1, default_root_dir='.', logger=None,
                 resume_from_checkpoint=None, callbacks=(), auto_lr_find=False, auto_scale_batch_size=False,
                 log_every_n_steps=50, trust_checkpoint=False, **_):
        # auto_lr_find / auto_scale_batch_size are inert in the reference as well (trainer.tune() is never
        # called, main.py:108-109)
        self.gpus = list(gpus) if isinstance(gpus, (list, tuple)) else ([gpus] if gpus is not None else [0])
        self.max_epochs, self.val_every = max_epochs, max(int(check_val_every_n_epoch), 1)
        self.root, self.logger, self.resume = default_root_dir, logger, resume_from_checkpoint
        self.ckpt = next((c for c in callbacks if isinstance(c, ModelCheckpoint)), None)
        self.log_every = log_every_n_steps
        self.trust_checkpoint = bool(trust_checkpoint)
        self.optimizer = None
        self.current_epoch, self.global_step = 0, 0

    def _device(self):
        rank, ws = sdist.init_from_env()
        if ws > 1:
            return torch.device('cuda', sdist.local_device_index())
        return torch.device('cuda', self.gpus[0])

    def save_checkpoint(self, model, path):
        rank, _ = sdist.world()
        if rank != 0:
            return
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        hp = {k: (_plain(vars(v)) if hasattr(v, '__dict__') else _plain(v)) for k, v in model.hparams.items()}
        hp = {k: v for k, v in hp.items() if v is not _DROP}
        torch.save({'state_dict': model.state_dict(), 'epoch': self.current_epoch, 'global_step': self.global_step,
                    'optimizer_states': [self.optimizer.state_dict()], 'hyper_parameters': hp}, path)

    def fit(self, model, datamodule):
        # this loop owns the process while it runs: let the model park long-lived objects outside the cyclic collector's reach
        # (lit_wrapper._maybe_freeze_gc) and give them back when the loop ends
        had = getattr(model, 'freeze_gc', None)
        if had is not None:
            model.freeze_gc = True
        try:
            return self._fit(model, datamodule)
        finally:
            if had is not None:
                model.freeze_gc = had
                getattr(model, 'unfreeze_gc', lambda: None)()

    def _fit(self, model, datamodule):
        device = self._device()
        model.to(device)
        model.trainer = self
        self.optimizer = _OptimizerProxy(model.configure_optimizers())
        if self.resume:
            ck = load_checkpoint(self.resume, map_location=device, trust=getattr(self, 'trust_checkpoint', False))
            model.load_state_dict(ck['state_dict'])
            if ck.get('optimizer_states'):
                self.optimizer.load_state_dict(ck['optimizer_states'][0])
            self.current_epoch = int(ck.get('epoch', -1)) + 1
            self.global_step = int(ck.get('global_step', 0))
        _, ws = sdist.world()
        if ws > 1:   # identical initial weights on every rank
            sdist.broadcast_([p.data for p in model.parameters()])
        train_loader = datamodule.train_dataloader()
        val_loader = datamodule.val_dataloader()
        ckpt_dir = (self.ckpt.dirpath if self.ckpt and self.ckpt.dirpath else os.path.join(self.root, 'checkpoints'))
        t0 = time.time()
        for epoch in range(self.current_epoch, self.max_epochs):
            self.current_epoch = epoch
            model.train()
            for i, batch in enumerate(train_loader):
                model.training_step(_to_device(batch, device), i)
                self.global_step += 1
                if self.logger is not None and self.global_step % self.log_every == 0:
                    self._flush(model)
            if val_loader is not None and (epoch + 1) % self.val_every == 0:
                model.eval()
                with torch.no_grad():
                    for i, batch in enumerate(val_loader):
                        model.validation_step(_to_device(batch, device), i)
                self._flush(model, extra={'epoch': epoch, 'wall_s': time.time() - t0})
            if self.ckpt is not None and (epoch + 1) % self.ckpt.period == 0:
                self.save_checkpoint(model, os.path.join(ckpt_dir, f'epoch={epoch}.ckpt'))
        return model

    def _flush(self, model, extra=None):
        if self.logger is None:
            return
        vals = {k: (float(v) if torch.is_tensor(v) else v) for k, v in model._logged.items()}
        vals.update(extra or {})
        self.logger.log_metrics(vals, self.global_step)
