"""``SingleVideoINN`` -- the training / validation / inference surface of the sin-inn path (drop-in for the
reference's lit_wrapper.py:12-138), running on the HIP kernels of ``sin-inn_amd``.

Same method names and step structure as the reference: forward pass HR -> (LR | z) with reconstruction + MMD +
latent terms, backward; reverse pass (LR | z) -> HR with reconstruction + MMD, backward; optional TCR iterations;
one optimizer step.  Differences, all deliberate: device-agnostic construction (no hard-coded 'cuda' strings),
``tcr_iters`` is cast to int (the reference feeds a float to range(), main.py:63 / lit_wrapper.py:63), latents are
drawn directly in the pixel-major layout the kernels use, Adam is the fused single-launch HIP optimizer, and ``infer``
clamps frames to [0,1] before the uint8 conversion (the reference's ToPILImage wraps out-of-range values around;
``opt.pixel_mode='wrap'`` restores that bit for bit).
"""
import logging
import os
import queue
import shutil
import subprocess as sp
import threading

import torch

import sin_inn_amd.lightning as pl
from sin_inn_amd import FusedAdam
from sin_inn_amd.functional import frames_to_u8

import loss
from archs import InvRescaleNet, UncondSRFlow
from tcr import TCR


_SECOND = {}


def _second_stream(device):
    key = str(device)
    if key not in _SECOND:
        from sin_inn_amd.modules import make_stream
        _SECOND[key] = make_stream(device, int(os.environ.get('SININN_PASS2_PRIO', '0')), 'second pass chain')
    return _SECOND[key]


def _latent(b, z_dims, h, w, device, temp=1.0):
    """z ~ N(0, temp^2), shape (b, z_dims, h, w), stored pixel-major (channels_last) like every other activation."""
    z = torch.randn(b, h, w, z_dims, device=device)
    if temp != 1.0:
        z = z * temp
    return z.permute(0, 3, 1, 2)


def _cat_channels(a, b):
    """torch.cat on dim 1 that keeps the pixel-major layout."""
    return torch.cat((a.permute(0, 2, 3, 1), b.permute(0, 2, 3, 1)), dim=3).permute(0, 3, 1, 2)


def _weighted_sum(like, *terms):
    """sum_i w_i * term_i() over the terms whose weight is non-zero.  The reference evaluates every term and multiplies by
    its weight, also when that is 0 (loss.mmd at the default flags, lit_wrapper.py:47,55); a term with
    weight 0 contributes exactly 0 to the loss and to every gradient, so it is not launched here, and a weight of 1 is
    not multiplied in (both bit-identical for FINITE values; known divergence: where the reference's skipped term is
    non-finite its 0 * inf = NaN poisons the logged loss and the gradients, here it does not).  With every weight 0 the
    result is a zero scalar on `like`'s device (no host/GPU hop in the step)."""
    total = None
    for weight, term in terms:
        if weight == 0:
            continue
        value = term()
        if weight != 1:
            value = weight * value
        total = value if total is None else total + value
    return total if total is not None else like.new_zeros(())


class _FrameWriter:
    """Writer side of SingleVideoINN.infer: PNG-encodes uint8 frames on a worker thread and writes them to
    ``save_images/out_{batch:04d}_{i:02d}.png`` or to the stdin of an ffmpeg process (reference lit_wrapper.py:96-103,117-124).
    The device -> host copies land in a ring of pinned buffers; the GPU never waits for the encoder unless all buffers are
    still being written out."""
    RING = 3

    def __init__(self, save_images=None, save_video=None):
        self.save_images, self.video = save_images, None
        if save_images:
            os.makedirs(save_images, exist_ok=True)
        elif save_video:
            if shutil.which('ffmpeg') is None:
                raise FileNotFoundError('ffmpeg is not on PATH (needed for save_video); use save_images instead')
            self.video = sp.Popen(['ffmpeg', '-framerate', '30', '-i', '-', '-c:v', 'libx264', '-preset', 'veryslow',
                                   '-crf', '18', '-y', save_video], stdin=sp.PIPE, stderr=sp.DEVNULL)
        self.free, self.work = queue.Queue(), queue.Queue()
        for _ in range(self.RING):
            self.free.put(None)                     # pinned buffers are allocated on first use (batch shape unknown yet)
        self.error = None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def submit(self, batch_index, frames_u8):
        """frames_u8: (n,H,W,C) uint8 on the device.  Returns as soon as the copy is queued."""
        if self.error is not None:
            raise self.error
        buf = self.free.get()
        if buf is None or buf.shape != frames_u8.shape:
            buf = torch.empty(frames_u8.shape, dtype=torch.uint8, pin_memory=True)
        buf.copy_(frames_u8, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        self.work.put((batch_index, buf, done, frames_u8))      # frames_u8 kept alive until the copy has finished

    def _run(self):
        from PIL import Image
        while True:
            item = self.work.get()
            if item is None:
                return
            bb, buf, done, _keep = item
            try:
                done.synchronize()
                for i, frame in enumerate(buf.numpy()):
                    im = Image.fromarray(frame)
                    if self.save_images:
                        im.save(os.path.join(self.save_images, f'out_{bb:04d}_{i:02d}.png'))
                    else:
                        im.save(self.video.stdin, 'PNG')
            except Exception as e:              # surfaced on the next submit() / close()
                self.error = e
            self.free.put(buf)

    def close(self):
        self.work.put(None)
        self.thread.join()
        if self.video is not None:
            self.video.stdin.close()
            self.video.communicate()
        if self.error is not None:
            raise self.error


class SingleVideoINN(pl.LightningModule):
    def __init__(self, c, h, w, opt):
        super().__init__()
        self.save_hyperparameters()
        self.opt = opt
        self.inn = {'SRF': UncondSRFlow, 'IRN': InvRescaleNet}[opt.architecture](c, h, w, opt)
        if getattr(opt, 'precision', 'fp32') != 'fp32':
            if not hasattr(self.inn, 'set_precision'):
                raise NotImplementedError(f'--precision {opt.precision} is implemented for the SRF architecture only')
            self.inn.set_precision(opt.precision)
        n_params = sum(p.numel() for p in self.inn.parameters())
        logging.info(f'Created model with {n_params / 1e6:.2f}M parameters. Using GPUs {opt.gpu_ids}')
        self.tcr = TCR(opt.rotation, opt.translation)
        self.automatic_optimization = False        # several backward() calls per step
        self.overlap_passes = True                 # forward / reverse pass on two HIP streams

    # ---- training --------------------------------------------------------------------------------
    # Host run-ahead is bounded to MAX_STEPS_IN_FLIGHT training steps: nothing in a step synchronises host and GPU any
    # more (frame indices go up pinned + non_blocking), so without this the host would queue steps as fast as Python allows;
    # every step in flight holds its own saved tensors (28 GB at 720p, -c 12) and the caching allocator then has to grow /
    # flush (a 10x slowdown was measured at BASELINE configs[4]).  One step of run-ahead is all the GPU needs to never idle;
    # batches of >= 8 M HR pixels (720p x 16: 170 ms of GPU work against 16 ms of host work per step) get none -- the
    # second step's saved tensors cost more (allocator growth, +3.5 %) than the ~1 ms start-up gap the run-ahead hides.
    MAX_STEPS_IN_FLIGHT = 2

    def _throttle(self, hr):
        ring = self.__dict__.setdefault('_step_events', [])
        depth = 1 if hr.numel() // hr.shape[1] >= (8 << 20) else self.MAX_STEPS_IN_FLIGHT
        while len(ring) >= depth:
            ring.pop(0).synchronize()                 # the step `depth` back has finished
        return ring

    # ---- hipGraph replay of the two pass chains -----------------------------------------------------------------------
    # One training step issues ~450 launches from Python autograd + the C++ block executors: 5.4-5.7 ms of host time, the
    # floor that the mixed-precision path at 256x256, IRN at small batches and -- under data parallel -- every rank's
    # host jitter in front of the collective sit on.  With `hip_graph` on (opt.hip_graph / SININN_GRAPH=1) the step's GPU
    # work from zeroing the gradients to the last weight-gradient kernel (both pass chains on their two streams + the
    # weight-gradient stream, forked and joined with events) is captured ONCE per (shapes, precision) after
    # GRAPH_WARMUP eager steps and replayed afterwards; the batch is copied into static input buffers in front of it.
    # Eager (never captured): the data-parallel all-reduce, the fused Adam launch and the pack refresh.  Not captured at
    # all: steps with TCR iterations or MMD terms (host-side randoms / a 16 x 16 Gram finish) -- they run eagerly as before.
    GRAPH_WARMUP = 3

    def _graph_wanted(self, batch):
        o = self.opt
        on = getattr(o, 'hip_graph', None)
        if on is None:
            on = os.environ.get('SININN_GRAPH', '0') == '1'
        return bool(on) and batch[0]['hr'].is_cuda and o.lambda_bwd_tcr == 0 and o.lambda_fwd_mmd == 0 and o.lambda_bwd_mmd == 0 \
            and not getattr(self, '_graph_broken', False)

    def _passes(self, hr, lr, batch, optim, join=True):
        """zero the gradients; forward pass + loss + backward, reverse pass + loss + backward (two streams when safe),
        TCR iterations.  Returns the logged loss (a device scalar)."""
        o = self.opt
        if join:
            optim.zero_grad()
        else:                                          # inside a capture: the weight-gradient stream was joined before it
            for g in optim.flat_grads_nojoin():
                g.zero_()
        b, _, h, w = lr.shape
        z = _latent(b, o.z_dims, h, w, hr.device)
        lr_z = _cat_channels(lr, z)

        # The two passes are independent until the optimizer step (the reverse pass starts from the ground-truth LR and a
        # fresh z, not from the forward output), so the reverse pass -- forward AND backward -- is queued on a second
        # HIP stream that only waits for the inputs: kernels of the two chains interleave on the GPU and fill each
        # other's prologue / epilogue / tail bubbles.  Weight gradients of both chains accumulate on ONE side stream
        # (sin_inn_amd.modules), so the += into the shared gradient buffers stays ordered.
        main = torch.cuda.current_stream()
        # only networks whose parameter gradients are all accumulated on the dedicated side stream may run two
        # backward chains at once (both the SRF graph and the IRN network do)
        concurrent = self.overlap_passes and getattr(self.inn, 'concurrent_passes_safe', False)
        second = _second_stream(hr.device) if concurrent else main
        if second is not main:
            # every packed-weight buffer must be complete on `main` before the fork: a cache miss inside one chain would
            # pack on that chain's stream while the other chain reads the same buffers (unsynchronised read)
            self.inn.prepare_packs()
            ready = torch.cuda.Event()
            ready.record(main)                       # inputs produced, gradients zeroed
            second.wait_event(ready)
        # Host order: reverse forward, forward forward, reverse backward, forward backward.  The GPU-side order inside each chain
        # and the order of the weight-gradient launches (reverse chain's first) are what they were; but when the queues are empty
        # -- the first step behind a device synchronisation -- the second chain's kernels now arrive after a quarter of the
        # step's host time instead of half, so the two chains overlap sooner.
        with torch.cuda.stream(second):
            # reverse pass: (LR | z) -> HR
            hr_hat = self.inn(lr_z, rev=True)
            bwd_loss = _weighted_sum(hr, (o.lambda_bwd_rec, lambda: loss.reconstruction(hr_hat, hr)),
                                     (o.lambda_bwd_mmd, lambda: loss.mmd(hr_hat, hr, rev=True)))

        # forward pass: HR -> (LR | z)
        lr_z_hat = self.inn(hr)
        fwd_loss = _weighted_sum(hr, (o.lambda_fwd_rec, lambda: loss.reconstruction(lr_z_hat[:, :o.lr_dims], lr)),
                                 (o.lambda_fwd_mmd, lambda: loss.mmd(lr_z_hat, lr_z)),
                                 (o.lambda_latent_nll, lambda: loss.latent_nll(lr_z_hat[:, o.lr_dims:])))
        with torch.cuda.stream(second):
            if bwd_loss.requires_grad:
                self.manual_backward(bwd_loss)
        if fwd_loss.requires_grad:
            self.manual_backward(fwd_loss)

        if second is not main:
            for t in (hr, lr, lr_z):
                t.record_stream(second)
            main.wait_stream(second)

        tcr_loss = 0
        if o.lambda_bwd_tcr > 0:
            # transformation consistency on the unsupervised pair: INN(warp(LR)) should equal warp(INN(LR))
            hr_u, lr_u = batch[1]['hr'], batch[1]['lr']
            iters = int(o.tcr_iters)
            for _ in range(iters):
                rand = torch.rand(b, 3)
                z = _latent(b, o.z_dims, h, w, hr_u.device)
                plain = _cat_channels(lr_u, z)
                warped = _cat_channels(self.tcr(lr_u, rand, scale=1 / o.scale), z)
                tcr_hr_hat = self.inn(warped, rev=True)
                hr_hat_tcr = self.tcr(self.inn(plain, rev=True), rand)
                tcr_loss = o.lambda_bwd_tcr / iters * loss.reconstruction(tcr_hr_hat, hr_hat_tcr)
                self.manual_backward(tcr_loss)
                tcr_loss = tcr_loss.detach()
        return fwd_loss.detach() + bwd_loss.detach() + tcr_loss

    def _graph_step(self, batch, optim):
        """Replay (or, once, capture) the pass chains for this batch; returns the loss scalar or None if the step must run
        eagerly (warm-up, capture refused by the runtime)."""
        from sin_inn_amd.modules import _PACK_REGISTRY, USE_SIDE_STREAM, USE_WINOGRAD, join_capturing_helpers, join_side_streams, \
            side_stream_if_any
        hr, lr = batch[0]['hr'], batch[0]['lr']
        o = self.opt
        # everything a capture bakes in besides the kernels' shapes: the loss weights (a zero weight removes a term's
        # launches, any other value is a kernel argument), which parameters are trained (NULL gradient pointers / skipped data
        # gradients), the conv algorithm switch, and the ADDRESSES of every packed-weight buffer -- the pack registry's generation
        # changes whenever one is allocated or dropped (Winograd switch, weights re-homed, load_state_dict onto new storage)
        trained = tuple(p.requires_grad for p in self.inn.parameters())
        key = (tuple(hr.shape), tuple(lr.shape), tuple(hr.stride()), tuple(lr.stride()), getattr(o, 'precision', 'fp32'),
               self.overlap_passes, USE_SIDE_STREAM[0], bool(USE_WINOGRAD[0]),
               (o.lambda_fwd_rec, o.lambda_fwd_mmd, o.lambda_latent_nll, o.lambda_bwd_rec, o.lambda_bwd_mmd, o.lambda_bwd_tcr, o.z_dims, o.lr_dims),
               hash(trained))
        graphs = self.__dict__.setdefault('_graphs', {})
        gen = _PACK_REGISTRY.generation
        for k in [k for k, v in graphs.items() if 'graph' in v and v['generation'] != gen]:
            del graphs[k]                              # a pack buffer it replays from may have been freed: capture again
        st = graphs.setdefault(key, {'seen': 0})
        st['seen'] += 1
        if 'graph' not in st:
            if st['seen'] <= self.GRAPH_WARMUP:
                return None                            # allocator, packs, lazily built maps and events settle eagerly first
            s_hr, s_lr = torch.empty_strided(hr.shape, hr.stride(), device=hr.device), \
                torch.empty_strided(lr.shape, lr.stride(), device=lr.device)
            s_hr.copy_(hr); s_lr.copy_(lr)
            join_side_streams()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, capture_error_mode='relaxed'):
                    out = self._passes(s_hr, s_lr, [{'hr': s_hr, 'lr': s_lr}], optim, join=False)
                    side = side_stream_if_any(hr.device)
                    if side is not None:               # the weight-gradient stream forked inside the capture: join it back
                        torch.cuda.current_stream().wait_stream(side)
                    out = out.clone()
                    # nothing may be left on a stream other than the capturing one when the capture ends ("the
                    # capture_end abort of round 3"): checked on the graph under construction, joined if found, and reported
                    loose = join_capturing_helpers()
                    if loose:
                        logging.warning('hipGraph capture: work was still unjoined on ' + ', '.join(loose) + ' (joined now)')
                        self.__dict__.setdefault('_capture_loose', []).extend(loose)
            except Exception as e:                     # noqa: BLE001 -- a runtime that refuses the capture must not stop training
                logging.warning(f'hipGraph capture refused ({type(e).__name__}: {e}); this model continues eagerly')
                self._graph_broken = True
                torch.cuda.synchronize()
                return None
            st.update(graph=g, hr=s_hr, lr=s_lr, loss=out, generation=_PACK_REGISTRY.generation)
            logging.info(f'captured the pass chains of a training step as one hipGraph for batch {tuple(hr.shape)}')
        else:
            st['hr'].copy_(hr, non_blocking=True); st['lr'].copy_(lr, non_blocking=True)
            join_side_streams()
        st['graph'].replay()
        side = side_stream_if_any(hr.device)
        if side is not None:
            # the eager weight-gradient stream saw none of the replayed kernels: order it behind the replay, so that whatever
            # orders itself behind IT (the data-parallel all-reduce, dist.allreduce_sum_(after=...)) is behind the gradients
            side.wait_stream(torch.cuda.current_stream())
        return st['loss']

    # ---- host jitter: the cyclic garbage collector --------------------------------------------------------------------
    # A full (generation-2) collection walks every container object of the process -- with torch imported that is 50 - 90 ms
    # of host time (rocprofv3 timeline: idle gaps of that length every ~10 - 30 steps), several times the two
    # steps of run-ahead the GPU has queued: the GPU drains and idles.  Whether one falls into a 20 - 30 step measurement is
    # what made the step time bimodal (9.8 vs 11.2 - 12.8 ms at configs[1]).  After GC_FREEZE_AFTER steps -- modules, packs,
    # streams, lazily built maps exist by then -- everything alive is moved to the permanent generation (gc.freeze): later
    # collections only look at what a step itself creates.  gc.freeze() is a PROCESS-WIDE interpreter setting, so the module
    # does not touch it on its own: the owner of the training loop opts in with `model.freeze_gc = True` -- the
    # Trainer of sin_inn_amd.lightning does for the duration of fit() and calls gc.unfreeze() when fit() returns, bench.py does
    # for its process -- or the environment does (SININN_GC_FREEZE=1; =0 forbids it whatever the owner says).
    GC_FREEZE_AFTER = 3
    _gc_frozen = [False]
    freeze_gc = False

    def _maybe_freeze_gc(self):
        n = self.__dict__.get('_steps_seen', 0) + 1
        self.__dict__['_steps_seen'] = n
        env = os.environ.get('SININN_GC_FREEZE')
        wanted = (self.freeze_gc or env == '1') and env != '0'
        if n == self.GC_FREEZE_AFTER and wanted and not SingleVideoINN._gc_frozen[0]:
            import gc
            gc.collect()
            gc.freeze()
            SingleVideoINN._gc_frozen[0] = True

    @staticmethod
    def unfreeze_gc():
        """Undo _maybe_freeze_gc (end of a training loop): the frozen objects return to the oldest generation and are
        collectable again."""
        if SingleVideoINN._gc_frozen[0]:
            import gc
            gc.unfreeze()
            SingleVideoINN._gc_frozen[0] = False

    def training_step(self, batch, batch_idx):
        self._maybe_freeze_gc()
        optim = self.optimizers()
        ring = self._throttle(batch[0]['hr']) if batch[0]['hr'].is_cuda else None
        total = None
        if self._graph_wanted(batch) and hasattr(optim, 'flat_grads_nojoin'):
            total = self._graph_step(batch, optim)
        if total is None:
            total = self._passes(batch[0]['hr'], batch[0]['lr'], batch, optim)
        optim.step()
        self.log('train', total)
        if ring is not None:
            ev = torch.cuda.Event()
            ev.record()                                # on the main stream, behind the optimiser step
            ring.append(ev)

    def validation_step(self, batch, batch_idx):
        o = self.opt
        hr, lr = batch['hr'], batch['lr']
        b, _, h, w = lr.shape
        lr_z = _cat_channels(lr, _latent(b, o.z_dims, h, w, hr.device))
        lr_z_hat = self.inn(hr)
        hr_hat = self.inn(lr_z, rev=True)
        self.log('lr_acc', loss.reconstruction(lr_z_hat[:, :o.lr_dims], lr))
        self.log('hr_acc', loss.reconstruction(hr_hat, hr))
        self.log('z_nll', loss.latent_nll(lr_z_hat[:, o.lr_dims:]))

    # ---- inference -------------------------------------------------------------------------------
    def infer(self, loader, opt, save_images=None, save_video=None):
        """Reverse pass over every LR window; frames go to PNG files or are piped to ffmpeg (reference :91-128).

        Pipeline: inverse pass (no autograd) -> on-device float->uint8 (sininn_frames_to_u8) -> asynchronous copy of the
        BYTES into one of three pinned host buffers -> a writer thread encodes PNGs and feeds the files / the ffmpeg pipe
        while the GPU already runs the next batch.  Pixel conversion: ``opt.pixel_mode`` 'clamp' (default: clamp to
        [0,1], then *255) or 'wrap' (the reference's ToPILImage = mul(255).byte(), which wraps out-of-range values
        around, lit_wrapper.py:94,120 -- kept selectable for bit-compatibility, not the default because it turns a
        slightly over-exposed pixel into a black one)."""
        self.inn.eval()
        device = self.device if self.device.type == 'cuda' else torch.device('cuda', opt.gpu_ids[0])
        self.inn.to(device)
        wrap = getattr(opt, 'pixel_mode', 'clamp') == 'wrap'
        writer = None
        if save_video or save_images:
            writer = _FrameWriter(save_images, save_video)
        outputs = []
        try:
            for bb, batch in enumerate(loader):
                lr = batch['lr'].to(device)
                b, _, h, w = lr.shape
                lr_z = _cat_channels(lr, _latent(b, opt.z_dims, h, w, device, temp=opt.temp))
                with torch.no_grad():
                    hr_hat = self.inn(lr_z, rev=True)
                if writer is not None:
                    writer.submit(bb, frames_to_u8(hr_hat, wrap=wrap))
                else:
                    outputs.append(hr_hat)
        finally:
            if writer is not None:
                writer.close()
        return outputs

    def configure_optimizers(self):
        return FusedAdam(self.parameters(), lr=self.opt.learning_rate, betas=tuple(self.opt.adam_betas),
                         weight_decay=self.opt.weight_decay)
