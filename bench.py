"""bench.py -- training frames/sec of the sin-inn INN training step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full SingleVideoINN.training_step (forward pass + loss + backward, reverse pass + loss + backward,
fused Adam) on a batch drawn by the frame-window sampler kernel from a synthetic uint8 clip that is already
resident in HBM.  Workload = BASELINE configs[1]: 8 GLOW blocks (-c 4 per level x 2 levels), 256x256x3 frames,
batch 16 per GPU, fp32 (f32 MFMA).  N>1: weak scaling, one RCCL all-reduce of the flat gradient per step.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the fused 3x3 coupling conv 256 -> 2*Co channels +
affine + log-det at level 0): `achieved` / `frac` count the FLOPs the matrix pipe EXECUTES (Winograd F(2x2,3x3) in fp32:
2.25x fewer than the direct convolution `achieved_algorithmic` credits) over the kernel's execution window (in-kernel
stamps == the duration a rocprofv3 kernel trace reports; the HIP-event bracket is given beside it); `cpu_baseline` times
the CPU oracle (torch-CPU restatement, "port") on every host core the process may use.
"""
import argparse
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_HBM_GBS = 8000.0


def make_opt(num_coupling, lr_window):
    o = types.SimpleNamespace(scale=4, num_coupling=num_coupling, lr_window=lr_window, architecture='SRF', gpu_ids=[0],
                              rotation=5.0, translation=5.0, tcr_iters=5, lambda_fwd_rec=1.0, lambda_fwd_mmd=0.0,
                              lambda_latent_nll=0.0, lambda_bwd_rec=1.0, lambda_bwd_mmd=0.0, lambda_bwd_tcr=0.0,
                              learning_rate=1e-4, adam_betas=[0.9, 0.99], weight_decay=1e-5, temp=0.8,
                              operation='train', fps=10, random_seed=0)
    o.lr_dims = (2 * lr_window + 1) * 4
    o.z_dims = 192 - o.lr_dims
    return o


class KernelTimer:
    """Live duration of every launch of the dominant kernel over the timed region, recorded by the C++ block executor
    (sininn_profile_begin / sininn_profile_end):
      * from inside the kernel: every block folds its entry / exit wall-clock time into two device words (atomic min / max)
        -> the kernel's own execution window, the quantity a rocprofv3 kernel trace reports;
      * HIP events on the launch stream around the launch: with the two pass chains and the weight-gradient stream in
        flight this bracket also contains the time the launch waits for the GPU behind other streams' kernels."""
    MAX_LAUNCHES = 4096

    def __init__(self, level_height, device):
        self.h, self.count, self.total_ms = level_height, 0, 0.0
        self.stamps = torch.empty((self.MAX_LAUNCHES, 2), dtype=torch.int64, device=device)
        self.stamp_ms = None

    def start(self):
        from sin_inn_amd import _lib
        self.stamps[:, 0] = torch.iinfo(torch.int64).max
        self.stamps[:, 1] = 0
        torch.cuda.synchronize()
        _lib.lib().sininn_profile_begin(self.h, self.stamps.data_ptr(), self.MAX_LAUNCHES)

    def stop(self):
        import ctypes as C
        from sin_inn_amd import _lib
        n, ms = C.c_int(0), C.c_float(0.0)
        _lib.check(_lib.lib().sininn_profile_end(C.byref(n), C.byref(ms)))
        self.count, self.total_ms = n.value, ms.value
        khz = _lib.lib().sininn_wall_clock_khz()
        st = self.stamps[:min(self.count, self.MAX_LAUNCHES)].cpu()
        ok = st[:, 1] > 0
        if khz > 0 and bool(ok.any()):
            self.stamp_ms = float((st[ok, 1] - st[ok, 0]).double().mean()) / khz

    def mean_ms(self):
        """Kernel execution window (in-kernel stamps); falls back to the event bracket."""
        if self.stamp_ms is not None:
            return self.stamp_ms
        return self.total_ms / self.count if self.count else None

    def mean_event_ms(self):
        return self.total_ms / self.count if self.count else None


def gpu_clocks(device_index=None):
    """Current shader / memory clock levels of the GPU(s) from sysfs (pp_dpm_sclk / pp_dpm_mclk: the line marked '*'), read
    as plain files -- no child process is started from a GPU-initialised process.  Logged beside the bench line so that a slow
    line can be told apart from a throttled or down-clocked box (round 2 saw configs[3] lines between 14.3 and 16.7 ms).
    The host's sysfs lists every GPU of the node (other tenants' too): with device_index the list is narrowed to the card whose
    PCI address is the torch device's."""
    import glob
    want = None
    if device_index is not None:
        try:
            pr = torch.cuda.get_device_properties(device_index)
            want = f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}'
        except (AttributeError, RuntimeError):
            want = None
    out, mine = [], []
    for dev in sorted(glob.glob('/sys/class/drm/card*/device')):
        rec = {}
        for key, name in (('sclk_mhz', 'pp_dpm_sclk'), ('mclk_mhz', 'pp_dpm_mclk')):
            try:
                for line in open(os.path.join(dev, name)):
                    if '*' in line:
                        rec[key] = int(''.join(ch for ch in line.split(':')[1] if ch.isdigit()))
            except (OSError, ValueError, IndexError):
                pass
        if rec:
            try:
                rec['busy_percent'] = int(open(os.path.join(dev, 'gpu_busy_percent')).read())
            except (OSError, ValueError):
                pass
            rec['card'] = os.path.basename(os.path.dirname(dev))
            out.append(rec)
            if want is not None and os.path.basename(os.path.realpath(dev)).lower().startswith(want):
                mine.append(rec)
    return mine or out or None


def host_cores():
    n = len(os.sched_getaffinity(0))
    try:                                       # cgroup v2 / v1 CPU quota
        if os.path.isfile('/sys/fs/cgroup/cpu.max'):
            q, p = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
            if q != 'max':
                n = min(n, max(1, int(float(q) / float(p) + 0.5)))
        elif os.path.isfile('/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
    except (OSError, ValueError):
        pass
    return int(os.environ.get('SININN_CPU_THREADS', n))


def cpu_baseline(args, opt, seconds_budget=30.0):
    """The CPU oracle's training step on the host cores (same architecture / sizes; bounded sample)."""
    from oracle import sininn_oracle as O
    torch.manual_seed(0)
    b = args.cpu_batch
    if getattr(args, 'arch', 'SRF') == 'IRN':
        ref = O.IRNOracle(3, opt.lr_dims, scale=4, num_coupling=args.num_coupling)
    else:
        ref = O.SRFlowOracle(3, args.height, args.width, scale=4, num_coupling=args.num_coupling)
    o = torch.optim.Adam(ref.parameters(), lr=1e-4, betas=(0.9, 0.99), weight_decay=1e-5)
    hr = torch.rand(b, 3, args.height, args.width)
    lr = torch.rand(b, opt.lr_dims, args.height // 8, args.width // 8)
    z = torch.randn(b, opt.z_dims, args.height // 8, args.width // 8)
    lam = dict(fwd_rec=1.0, fwd_mmd=0.0, latent_nll=0.0, bwd_rec=1.0, bwd_mmd=0.0)
    # every host core this process may use: the affinity mask, capped by the container's CPU quota when there is one (a
    # box exposes all of the node's cores in the mask but schedules its CPU share); SININN_CPU_THREADS overrides
    cores = host_cores()
    torch.set_num_threads(cores)
    tw = time.time()
    O.training_step(ref, hr, lr, z, lam, opt.lr_dims); o.step()          # warm-up
    tw = time.time() - tw
    t0, n = time.time(), 0
    while tw < seconds_budget and (n < 1 or (time.time() - t0 < seconds_budget * 0.5 and n < 3)):
        O.training_step(ref, hr, lr, z, lam, opt.lr_dims); o.step()
        n += 1
    dt = (time.time() - t0) / n if n else tw
    return {'value': b / dt, 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'host_cores': {'used': torch.get_num_threads(), 'affinity_mask': len(os.sched_getaffinity(0)), 'node': os.cpu_count()},
            'sample': f'{n} timed + 1 warm-up training steps of the torch-CPU oracle (fp32), batch {b}'
                      + ('' if b == args.batch else f' (frames/s = batch / step time, per-frame normalised from batch {b}, not the benchmark batch {args.batch})')
                      + f', {args.width}x{args.height}, -c {args.num_coupling}, {getattr(args, "arch", "SRF")}, '
                      f'{torch.get_num_threads()} threads = every core this process may use'}


_JSON_FD = [None]


def claim_stdout():
    """fd 1 carries the JSON line and nothing else.  Native libraries print there too (gloo: "[Gloo] Rank 0 is connected to ..."),
    so from here on everything written to fd 1 lands on stderr and emit() alone writes to the real stdout."""
    sys.stdout.flush()
    _JSON_FD[0] = os.dup(1)
    os.dup2(2, 1)


def emit(line):
    sys.stdout.flush()
    os.write(1 if _JSON_FD[0] is None else _JSON_FD[0], (line + '\n').encode())   # one write: the ranks share the launcher's pipe


def self_launch(n):
    """Run this script under torch.distributed.run with n ranks (one per GPU) and relay rank 0's JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:                 # rank 0 prints exactly one JSON line on stdout; stderr passes through
        if line.startswith('{'):
            emit(line.rstrip('\n'))
        else:
            sys.stderr.write(line)
    return proc.wait()


def rehearse(args, rank, ws):
    """The distributed skeleton of the benchmark without the GPU work (runs on the CPU-only build container)."""
    from sin_inn_amd import dist as sdist
    flat = torch.full((1 << 16,), float(rank + 1))
    if ws > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sdist.allreduce_sum_([flat])
        flat.mul_(1.0 / ws)
    if ws > 1:
        torch.distributed.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if ws > 1:
        torch.distributed.all_reduce(dt, op=torch.distributed.ReduceOp.MAX)
    assert abs(float(flat[0]) - (ws + 1) / 2.0) < 1e-5
    ranks = rank_table(None)
    if rank == 0:
        line = json.dumps({'metric': 'training frames/sec at 256x256 bs=16', 'value': None, 'unit': 'frames/s', 'n_gpus': ws,
                          'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': float(dt) / args.steps * 1e3,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
                          'data': 'synthetic', 'rehearsal': True, 'distributed': distributed_record(ranks, None, float(dt) / args.steps * 1e3),
                          'config': {'workload': 'REHEARSAL of the multi-rank plumbing on CPU (gloo): no kernels run',
                                     'global_batch': ws * args.batch, 'parallelism': f'dp{ws}'}})
        emit(line)
    return 0


def rank_table(device_index):
    """One record per rank (gathered on every rank): which process drove which GPU.  Lets the reader of an N-GPU line check that
    N distinct devices took part (PCI address), not N ranks on one card."""
    import torch.distributed as dist
    rec = {'rank': int(os.environ.get('RANK', '0')), 'local_rank': int(os.environ.get('LOCAL_RANK', '0')), 'pid': os.getpid(),
           'host': os.uname().nodename}
    if device_index is not None:
        pr = torch.cuda.get_device_properties(device_index)
        rec.update(device=f'cuda:{device_index}', name=pr.name,
                   pci=(f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}' if hasattr(pr, 'pci_bus_id') else None))
    else:
        rec.update(device='cpu')
    if not (dist.is_available() and dist.is_initialized()):
        return [rec]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, rec)
    return out


def distributed_record(ranks, allreduce_ms, max_step_ms):
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized()
    return {'world_size': dist.get_world_size() if on else 1, 'backend': dist.get_backend() if on else None, 'ranks': ranks,
            'distinct_devices': len({(r.get('host'), r.get('pci') or r.get('device')) for r in ranks}),
            'allreduce_ms_per_step': allreduce_ms,
            'allreduce_note': 'HIP events per step, rank 0: from the last weight-gradient kernel (the stream the collective is ordered '
                              'behind) to the reduced flat gradient being visible to the stream that launches Adam',
            'ms_per_step_max_over_ranks': max_step_ms,
            'collective': 'one all-reduce (sum) of the flat fp32 gradient buffer per step; 1/world folded into the Adam launch'}


CONFIGS = {
    # BASELINE.json configs[i] -> workload presets (height, width, -c, precision, per-GPU batch)
    1: dict(height=256, width=256, num_coupling=4, precision='fp32', batch=16,
            name='BASELINE configs[1]: 8-block INN (SRF, -c 4 x 2 levels), 256x256x3, fp32'),
    2: dict(height=256, width=256, num_coupling=4, precision='fp32', batch=16,
            name='BASELINE configs[2]: the configs[1] clip at global batch 128 = 16 per GPU x 8 GPUs (run with --gpus 8; with fewer '
                 'GPUs the per-GPU work is the same and the global batch is 16 x N), one RCCL all-reduce of the flat gradient per step'),
    3: dict(height=512, width=512, num_coupling=4, precision='bf16', batch=16,
            name='BASELINE configs[3] shape: INN at 512x512x3 (SRF, -c 4 x 2 levels), bf16 conv subnets / fp32 flow'),
    4: dict(height=720, width=1280, num_coupling=12, precision='bf16', batch=16,
            name='BASELINE configs[4] shape: 720p (1280x720x3), 12 GLOW blocks per level (-c 12), bf16 conv subnets / fp32 flow'),
}
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak (the 5 PF headline includes 2:1 sparsity)

CLASS_NAMES = ['conv1 forward (+ReLU)', 'conv2 forward + coupling + log-det', 'data gradient of conv2 (+ReLU mask)',
               'data gradient of conv1 (+skip grad, +fused coupling backward)', 'weight gradients (+slab reduce)',
               'coupling backward tail (HBM-bound, no FLOPs counted)']


IRN_CLASS_NAMES = ['DenseBlock conv1-4 forward (+LeakyReLU, 32 output channels each)', 'DenseBlock conv5 forward (+ fused add / affine tail)',
                   'data gradient of conv5', 'data gradients of conv1-4 (accumulating into the feature-gradient buffer)',
                   'the five weight gradients of a DenseBlock (one grouped launch + ordered slab reduce)',
                   'elementwise: LeakyReLU backward, affine-tail backward, channel copies (HBM-bound, no FLOPs counted)']


def class_roofline(precision, arch='SRF'):
    """roofline.classes from the executor's per-launch HIP-event brackets (single-stream steps): algorithmic TF/s per kernel
    class, its fraction of the dtype's dense MFMA peak, and -- for the Winograd classes of the fp32 path -- the fraction of
    the f32 matrix pipe the EXECUTED FLOPs occupy (Winograd F(2x2,3x3) executes 2.25x fewer than it is credited with)."""
    import ctypes as C
    from sin_inn_amd import _lib
    n = 12
    ms, fl, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_int * n)()
    _lib.check(_lib.lib().sininn_profile_classes_end(n, ms, fl, cnt))
    by = (C.c_double * n)()
    _lib.check(_lib.lib().sininn_profile_classes_bytes(n, by))
    out = []
    for i in range(n):
        if cnt[i] == 0:
            continue
        k = 1 if i >= 6 else 3
        cls = i % 6
        # fp32 path: f32 matrix pipe everywhere; bf16 path: every conv class AND the weight gradients (transposing
        # bf16-MFMA kernel, fp32 accumulation) run on the bf16 pipe
        peak = PEAK_F32_MFMA_TFLOPS if precision == 'fp32' else PEAK_BF16_MFMA_TFLOPS
        rec = {'class': f'{k}x{k} {(IRN_CLASS_NAMES if arch == "IRN" else CLASS_NAMES)[cls]}', 'launches': cnt[i], 'ms': ms[i]}
        if fl[i] > 0:
            tf = fl[i] / (ms[i] * 1e-3) / 1e12
            wino = k == 3 and precision == 'fp32'
            ex = tf / 2.25 if wino else tf           # what the matrix pipe executes (Winograd F(2x2,3x3): 16 instead of 36 multiplies)
            rec.update(alg_tflops=tf, executed_tflops=ex, peak_tflops=peak, frac=ex / peak, frac_algorithmic=tf / peak, bound='mfma')
            if wino:
                rec['note'] = 'Winograd: frac counts the EXECUTED MFMA FLOPs (2.25x fewer than the algorithmic direct-conv count)'
        else:
            rec['bound'] = 'hbm'
        if by[i] > 0:      # the fused 1x1 launches report their algorithmic HBM bytes: the second roof of a class that sits at the ridge
            gbs = by[i] / (ms[i] * 1e-3) / 1e9
            rec['hbm'] = {'alg_bytes': by[i], 'achieved_gbs': gbs, 'peak_gbs': PEAK_HBM_GBS, 'frac': gbs / PEAK_HBM_GBS,
                          'note': 'algorithmic bytes (x / dr / side inputs / outputs / stored hidden tensor / slabs) over the class time'}
        out.append(rec)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=int, default=1, choices=sorted(CONFIGS), help='BASELINE.json configs[] index: 1 = the headline '
                    '(256x256 bs 16 fp32), 2 = the same per GPU under data parallel (bs 128 on 8 GPUs), 3 = 512x512 bf16, 4 = 720p -c 12 bf16')
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: the config preset)')
    ap.add_argument('--size', type=int, default=None, help='square frame size override')
    ap.add_argument('--height', type=int, default=None)
    ap.add_argument('--width', type=int, default=None)
    ap.add_argument('--num-coupling', type=int, default=None, help='GLOW blocks per level (2 levels): 4 -> 8-block INN')
    ap.add_argument('--precision', choices=['fp32', 'bf16'], default=None)
    ap.add_argument('--arch', choices=['SRF', 'IRN'], default='SRF', help='SRF = the GLOW-block INN of the headline; IRN = the '
                    'reference\'s second architecture (-a IRN: Haar + DenseBlock couplings, archs.py:135-233), fp32 only')
    ap.add_argument('--lr-window', type=int, default=10)
    ap.add_argument('--frames', type=int, default=64)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=None)
    ap.add_argument('--conv-cfg', type=int, default=0, help='diagnostic (sininn_conv_test_hooks force_cfg): 1 / 2 pin 32- / 64-column Winograd blocks')
    ap.add_argument('--wgrad16', type=int, default=0, help='diagnostic (sininn_wgrad_test_hooks): bit0 16x16x4 tiles, bit1 no Winograd wgrad, bit2 8-row tiles, bit3 8-wave k-split blocks, bit4 per-conv launches, bit5 per-half groups')
    ap.add_argument('--rehearse', action='store_true', help='CPU rehearsal of the multi-rank plumbing (gloo): rendezvous, '
                    'flat-gradient all-reduce, barrier + max-over-ranks timing, rank-0 JSON line; no kernels run, value is null')
    ap.add_argument('--with-flow', action='store_true', help='BASELINE configs[3] as named ("pair_flow warp + INN at 512x512, bf16"): '
                    'every step first warps the neighbouring frame of each sample by a resident flow field and takes the photometric '
                    'metric against the sample (flow_warp_l1, bf16 images / fp32 flow, video-interpolation/trainer.py:61-62) and '
                    'its gradient w.r.t. the flow, then runs the INN training step on the batch')
    ap.add_argument('--with-tcr', action='store_true', help='lambda_bwd_tcr = 1: every step also runs the transformation-consistency branch '
                    '(reference lit_wrapper.py:58-72: per iteration two more inverse passes, two affine warps, an MSE and a backward)')
    ap.add_argument('--tcr-iters', type=int, default=1, help='TCR samples per step with --with-tcr (reference default: 5)')
    ap.add_argument('--graph', choices=['on', 'off'], default='off', help='replay the pass chains of a step as one hipGraph '
                    '(lit_wrapper: captured after 3 eager steps; a refused capture falls back to eager launches)')
    ap.add_argument('--no-overlap', action='store_true', help='diagnostic: single stream (no pass / wgrad overlap)')
    ap.add_argument('--overlap', choices=['auto', 'full', 'wgrad', 'none'], default='auto',
                    help='full: forward / reverse pass chains on two streams + weight gradients on a third (default below 2 M level-0 '
                         'pixels per batch); wgrad: one chain + the weight-gradient stream; none == --no-overlap')
    args = ap.parse_args()
    claim_stdout()
    preset = CONFIGS[args.config]
    if args.size is not None:
        args.height = args.width = args.size
    for k in ('height', 'width', 'num_coupling', 'precision', 'batch'):
        if getattr(args, k) is None:
            setattr(args, k, preset[k])
    if args.cpu_batch is None:
        # the benchmark's own batch where a CPU step is about a second (configs[1]); the bigger configs: a bounded sample
        args.cpu_batch = args.batch if args.height * args.width <= 256 * 256 else 1
    custom = any(getattr(args, k) != preset[k] for k in ('height', 'width', 'num_coupling', 'precision', 'batch'))

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing in this process has touched the GPU yet (torch
        # is imported, no device call), and the ranks are fresh child processes -- never an exec of a GPU-initialised one.
        sys.exit(self_launch(args.gpus))

    import sin_inn_amd
    from sin_inn_amd import dist as sdist
    rank, ws = sdist.init_from_env('gloo' if args.rehearse else None)
    assert ws == args.gpus or (ws == 1 and args.gpus == 1), f'--gpus {args.gpus} but WORLD_SIZE={ws}'
    if args.rehearse:
        return rehearse(args, rank, ws)
    local = sdist.local_device_index()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    import lit_wrapper
    from data import FrameStore
    from sin_inn_amd.functional import sample_windows

    from sin_inn_amd import _lib as _l, modules as _m
    _l.lib().sininn_wgrad_test_hooks(args.wgrad16)
    _l.lib().sininn_conv_test_hooks(args.conv_cfg, 0)
    if args.no_overlap:
        _m.USE_SIDE_STREAM[0] = False
    opt = make_opt(args.num_coupling, args.lr_window)
    opt.precision = args.precision
    opt.architecture = args.arch
    opt.hip_graph = args.graph == 'on'
    if args.with_tcr:
        opt.lambda_bwd_tcr, opt.tcr_iters = 1.0, args.tcr_iters
    if opt.hip_graph:
        args.warmup = max(args.warmup, 5)        # 3 eager steps + the capturing one + a replay before the timed region
    irn = args.arch == 'IRN'
    if irn:
        assert args.precision == 'fp32', 'the IRN path is fp32 only'
        custom = True
    torch.manual_seed(0)                                   # identical random-init weights on every rank
    model = lit_wrapper.SingleVideoINN(3, args.height, args.width, opt).to(dev)
    model.freeze_gc = True                                 # this process is the benchmark's alone (lit_wrapper: opt-in)
    if args.no_overlap or args.overlap == 'none':
        args.no_overlap = True
        _m.USE_SIDE_STREAM[0] = False
        model.overlap_passes = False
    elif args.overlap == 'wgrad':
        model.overlap_passes = False
    elif args.overlap == 'full':
        model.overlap_passes = True
    optim = model.attach_optimizer()
    store = FrameStore.synthetic(args.frames, args.height, args.width).to(dev)   # clip resident in HBM before timing
    lo, hi = args.lr_window, args.frames - args.lr_window
    gen = torch.Generator().manual_seed(100 + rank)        # each rank draws its own frames (data-parallel shard)

    lh, lw = args.height // 4, args.width // 4             # level-0 resolution
    b, m0 = args.batch, args.batch * lh * lw
    co0 = 24

    timer = KernelTimer(lh, dev)      # forward 3x3 coupling conv (256 -> 2*24 columns) at level-0 resolution

    # frame indices of every step drawn up front and resident in HBM like the clip itself: a per-step pageable H2D copy is
    # synchronous -- it would make the host wait for the previous step's last kernel before it may enqueue the next step
    n_calls = args.warmup + args.steps + 8
    idx_all = torch.randint(lo, hi, (n_calls, b), generator=gen).to(device=dev, dtype=torch.int32)
    calls = [0]

    flow_ev = []            # (start, mid, end) events around the warp's two kernels, single-stream phase only
    flow_field = None
    if args.with_flow:
        from sin_inn_amd.functional import flow_warp_l1, sample_pairs
        # a smooth synthetic flow (what a flow network emits): low-frequency sines, up to ~4 pixels
        yy, xx = torch.meshgrid(torch.arange(args.height, device=dev, dtype=torch.float32),
                                torch.arange(args.width, device=dev, dtype=torch.float32), indexing='ij')
        fx = 3.0 * torch.sin(yy / 37.0) + 1.5 * torch.cos(xx / 23.0)
        fy = 2.5 * torch.cos(yy / 29.0 + xx / 41.0)
        flow_field = torch.stack((fx, fy)).unsqueeze(0).repeat(b, 1, 1, 1).contiguous().requires_grad_(True)
        img_dtype = torch.bfloat16 if args.precision == 'bf16' else torch.float32

    def flow_part(hr, idx, record):
        # pair_flow: the neighbouring frame of every sample, warped onto the sample by the flow; metric + d metric / d flow
        tgt, img = sample_pairs(store.hr, idx, gap=1, dtype=img_dtype)     # (sample, its neighbour) planar, straight from the u8 clip
        flow_field.grad = None
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if record else None
        if record:
            ev[0].record()
        warped, metric = flow_warp_l1(img, flow_field, tgt)
        if record:
            ev[1].record()
        metric.backward(torch.full_like(metric, 1.0 / metric.numel()))            # d mean(metric) / d flow (the image is data)
        if record:
            ev[2].record()
            flow_ev.append(ev)

    record_flow = [False]

    def step():
        idx = idx_all[calls[0] % n_calls]
        calls[0] += 1
        hr, lr = sample_windows(store.hr, store.lr, idx, args.lr_window)
        if args.with_flow:
            flow_part(hr, idx, record_flow[0])
        batch = {'hr': hr, 'lr': lr}
        model.training_step([batch, batch], 0)

    def barrier():
        if ws > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # diagnostic: run the steps on a stream of explicit priority instead of the default stream (SININN_MAIN_PRIO)
    main_prio = os.environ.get('SININN_MAIN_PRIO')
    if main_prio is not None:
        work_stream = _m.make_stream(dev, int(main_prio))
        work_stream.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(work_stream)
    for _ in range(args.warmup):
        step()
    barrier()
    print(f'[bench] rank {rank}: warm-up done', file=sys.stderr, flush=True)
    clocks_before = gpu_clocks(local)
    ranks = rank_table(local)
    assert len(ranks) == ws == args.gpus, f'--gpus {args.gpus}: {len(ranks)} ranks answered, world size {ws}'
    sdist.TIMING[0] = [] if ws > 1 else None
    timer.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_issue = time.perf_counter() - t0      # host time to enqueue the steps (GPU still running)
    barrier()
    dt = time.perf_counter() - t0
    clocks_after = gpu_clocks(local)
    timer.stop()
    allreduce_ms = None
    if sdist.TIMING[0]:
        allreduce_ms = sum(a.elapsed_time(b) for a, b in sdist.TIMING[0]) / len(sdist.TIMING[0])
    sdist.TIMING[0] = None
    if ws > 1:
        t = torch.tensor([dt], device=dev if torch.distributed.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    # The roofline needs kernel durations with the chip to itself: in the timed region above the blocks of a kernel
    # interleave with the other pass chain's and the weight-gradient stream's kernels, and a rocprofv3 trace of this
    # command serialises most of that overlap away.  So the same step is run a few more times on ONE stream (not part of
    # `value`): the dominant kernel is timed there, and so is every kernel class of the block executor (roofline.classes).
    iso, classes, iso_ms = None, None, None
    if True:                                                # every rank runs them: the optimiser step holds a collective
        _m.USE_SIDE_STREAM[0] = False
        model.overlap_passes = False
        model.opt.hip_graph = False                         # the per-class event brackets live in the executors' launch path
        iso = KernelTimer(lh, dev)
        step()                                              # settle on one stream
        barrier()
        iso.start()
        _l.lib().sininn_profile_classes_begin()
        n_iso = min(3, args.steps)
        record_flow[0] = True
        t1 = time.perf_counter()
        for _ in range(n_iso):
            step()
        torch.cuda.synchronize()
        record_flow[0] = False
        iso_ms = (time.perf_counter() - t1) / n_iso * 1e3
        iso.stop()
        classes = class_roofline(args.precision, args.arch)
        for c in classes:
            c['ms_per_step'] = c.pop('ms') / n_iso
            if 'hbm' in c:
                c['hbm']['alg_bytes_per_step'] = c['hbm'].pop('alg_bytes') / n_iso
            c['launches_per_step'] = c.pop('launches') // n_iso
        if flow_ev:
            # HBM rows of the warp (north_star's second fused kernel).  Algorithmic bytes per pixel (e = bytes of an image
            # element): forward reads img 3e + flow 8 + target 3e, writes warped 3e + metric 4; the flow-only backward reads
            # img 3e (4 taps, neighbours from cache) + flow 8 + target 3e + warped 3e + gmetric 4 and writes gflow 8.
            e = 2 if args.precision == 'bf16' else 4
            px = float(b * args.height * args.width)
            f_ms = sum(ev[0].elapsed_time(ev[1]) for ev in flow_ev) / len(flow_ev)
            b_ms = sum(ev[1].elapsed_time(ev[2]) for ev in flow_ev) / len(flow_ev)
            for name, ms_, byts in (('flow warp + photometric L1 forward (flow_warp_l1_kernel)', f_ms, px * (9 * e + 12)),
                                    ('flow warp backward w.r.t. the flow (flow_warp_l1_bwd_kernel; + the 1/N fill)', b_ms, px * (9 * e + 20))):
                gbs = byts / (ms_ * 1e-3) / 1e9
                classes.append({'class': name, 'ms_per_step': ms_, 'launches_per_step': 1, 'bound': 'hbm', 'alg_bytes': byts,
                                'achieved_gbs': gbs, 'peak_gbs': PEAK_HBM_GBS, 'frac': gbs / PEAK_HBM_GBS,
                                'note': f'HIP events on the stream, image element {e} B'})

    if rank != 0:
        return
    print(f'[bench] timed region {dt:.3f}s for {args.steps} steps (host enqueue time {t_issue:.3f}s)', file=sys.stderr, flush=True)
    ms_per_step = dt / args.steps * 1e3
    value = ws * b * args.steps / dt
    # roofline of the dominant kernel: algorithmic FLOPs (2 * pixels * 9*256 * 2*Co) per launch
    # Per-launch duration of the dominant kernel: its EXECUTION WINDOW (first block's entry to last block's exit, stamped from
    # inside the kernel, sininn_conv_args.stamp) over the single-stream launches -- the quantity a rocprofv3 kernel trace
    # reports for the same kernel + grid (profiles/*_single_stream_by_grid.csv; the window starts a few us after the
    # dispatch, so the trace reads ~5 % longer).  The HIP-event bracket on the launch stream is reported next to it: it also
    # contains the two event packets' dispatch gaps (~12 us), i.e. time in which the GPU is free for other streams' kernels.
    tm = iso or timer
    kms_events = tm.mean_event_ms()
    kms = tm.stamp_ms if tm.stamp_ms is not None else kms_events
    kms_src = ('in-kernel execution window (s_memrealtime stamps), single-stream steps run right after the timed region'
               if tm.stamp_ms is not None else 'HIP events on the launch stream (no in-kernel stamps in this kernel), single-stream steps '
               'run right after the timed region')
    flops = 2.0 * m0 * 9 * 256 * (2 * co0)
    roof = None
    bf16 = args.precision == 'bf16'
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, 'profiles', 'traffic_dominant.json')   # PMC pass is a separate rocprofv3 run (committed)
    if os.path.isfile(tpath) and args.config in (1, 2) and not custom:
        tj = json.load(open(tpath))
        traffic, traffic_src = tj.get('hbm_bytes_per_launch'), tj.get('source')
    # rocprofv3's duration of the same kernels from the committed trace of this build (profiles/roofline_kernels.json, written by
    # tools/refresh_profiles.sh from profiles/*_single_stream_by_grid.csv with the commit it was measured at): `frac` is computed
    # from THAT when it is there -- a number the reader can re-derive from profiles/ -- and the live stamp window is kept beside it
    rk = None
    rpath = os.path.join(ROOT, 'profiles', 'roofline_kernels.json')
    if os.path.isfile(rpath) and args.config in (1, 2) and not custom and not bf16 and not irn:
        rk = json.load(open(rpath))
    kms_live = kms
    if rk and rk.get('named'):
        kms = rk['named']['avg_us'] * 1e-3
    if kms:
        ach = flops / (kms * 1e-3) / 1e12
        hid_bytes = 2.0 if bf16 else 4.0
        abytes = hid_bytes * m0 * 256 + 4.0 * m0 * 3 * co0 + hid_bytes * 9 * 256 * 2 * co0
        kernel = ('conv_bf16_kernel<3,32,1,8,true> (fused 3x3 coupling conv 256->48 + affine + log-det, level 0; direct implicit GEMM on '
                  'v_mfma_f32_32x32x16_bf16, bf16 hidden tensor in, fp32 flow out)') if bf16 else \
                 ('wino_kernel<2,8,2> (fused 3x3 coupling conv 256->48 + affine + log-det, level 0; Winograd F(2x2,3x3): executes '
                  '2.25x fewer MFMA FLOPs than the algorithmic direct-conv count used here)')
        # fp32: the kernel is Winograd F(2x2,3x3) -- the matrix pipe executes 16 multiplies per 2x2 outputs where the direct
        # convolution the algorithmic count credits needs 36.  A roofline fraction has to be about work the pipe EXECUTES, so
        # `achieved` / `frac` are the executed rate; the algorithmic (direct-convolution) rate is kept beside it.
        executed = ach if bf16 else ach / 2.25
        roof = {'bound': 'mfma', 'achieved': executed, 'peak': peak, 'unit': 'TFLOP/s',
                'frac': executed / peak, 'traffic': traffic, 'traffic_source': traffic_src,
                'kernel': kernel,
                'achieved_algorithmic': ach, 'frac_algorithmic': ach / peak,
                'launches_timed': tm.count, 'avg_ms': kms,
                'avg_ms_source': (f"rocprofv3 --kernel-trace of `bench.py --no-overlap` on commit {rk.get('commit')}: {rk['named']['kernel']} "
                                  f"grid {rk['named']['grid']}, {rk['named']['calls']} calls ({rk.get('source')})") if rk and rk.get('named') else kms_src,
                'avg_ms_rocprof': rk['named']['avg_us'] * 1e-3 if rk and rk.get('named') else None,
                'avg_ms_execution_window': kms_live, 'avg_ms_execution_window_source': kms_src,
                'avg_ms_hip_events': kms_events,
                'timed_region': {'launches': timer.count, 'avg_ms_execution_window': timer.stamp_ms,
                                 'avg_ms_hip_events': timer.mean_event_ms(),
                                 'note': 'three streams in flight: the window contains other kernels\' blocks'},
                'alg_flops_per_launch': flops, 'executed_flops_per_launch': flops if bf16 else flops / 2.25,
                'alg_bytes_per_launch': abytes}
        if not bf16:
            roof['limiter'] = ('nearest roof is the f32 matrix pipe (168 algorithmic FLOP/B against a ridge of 20), but the pipe is '
                               f'only {executed / peak:.0%} busy: vector-ALU and LDS instructions do not hide behind an MFMA on this chip '
                               '(tools/mfma_valu.hip: +2.5 / +4.6 clocks each per 32-clock MFMA), so the per-lane Winograd transforms and '
                               'the fragment reads cost issue time in the K loop (~0.83 of the pipe there), and the block-level ablation '
                               '(profiles/r03_wino_fwd_ablation.log) puts the rest in operand staging, the chunk barrier and the epilogue')
        else:
            roof['limiter'] = ('nearest roof is the bf16 matrix pipe; the MFMA loop is LDS-bandwidth-bound (one 16-byte fragment read per '
                               '32x32x16 MFMA and wave = the 128 B/clk of a CU), and phase stamps (DESIGN 6) put two thirds of the block time '
                               'in operand staging and the epilogue store drain around it')
        # the same launch against the HBM roof (north_star quotes an HBM fraction): algorithmic bytes / time / 8 TB/s
        roof['hbm'] = {'achieved': abytes / (kms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                       'frac': abytes / (kms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                       'note': 'FLOP/B far above the ridge: the kernel is matrix-pipe-bound, this fraction is low by nature'}
        roof['classes'] = classes
        roof['single_stream_ms_per_step'] = iso_ms
        if not bf16:
            # the kernel that is dominant BY TIME: the grouped Winograd weight gradient of a 3x3 block (four convs, one launch)
            wg = next((c for c in classes if c['class'].startswith('3x3 weight gradients')), None)
            wflops = 2.0 * m0 * 9 * (24 * 256 + 256 * 48) * 2          # per block, either level: SURVEY 8d (43.49 GFLOP)
            wms_live = wg['ms_per_step'] / max(wg['launches_per_step'], 1) if wg else None
            wms_prof = rk['by_time']['avg_us'] * 1e-3 if rk and rk.get('by_time') else None
            wms = wms_prof or wms_live
            if wms:
                roof['dominant_by_time'] = {
                    'kernel': 'wgrad_wino_group_kernel<4,2> (the four 3x3 weight gradients of a GLOW block in one launch, Winograd F(2x2,3x3) '
                              'on v_mfma_f32_16x16x4_f32, + ordered slab reduce)',
                    'bound': 'mfma', 'peak': peak, 'unit': 'TFLOP/s', 'alg_flops_per_launch': wflops, 'executed_flops_per_launch': wflops / 2.25,
                    'avg_ms': wms, 'avg_ms_rocprof': wms_prof, 'avg_ms_live_event_bracket_incl_reduce': wms_live,
                    'avg_ms_source': (f"rocprofv3 --kernel-trace on commit {rk.get('commit')}: grid {rk['by_time']['grid']}, "
                                      f"{rk['by_time']['calls']} calls, {rk['by_time']['percent']} % of the GPU time of a single-stream step "
                                      f"({rk.get('source')})") if wms_prof else 'HIP-event bracket of the class (kernel + reduce), single-stream steps',
                    'achieved': wflops / 2.25 / (wms * 1e-3) / 1e12, 'frac': wflops / 2.25 / (wms * 1e-3) / 1e12 / peak,
                    'achieved_algorithmic': wflops / (wms * 1e-3) / 1e12}
        # whole-step algorithmic rate (train step = 6 x forward FLOPs)
        fwd = 0.0
        for lvl, (mm, cc) in enumerate(((m0, 48), (m0 // 4, 192))):
            for blk in range(args.num_coupling):
                kk = 9 if blk % 2 == 0 else 1
                fwd += 2.0 * mm * kk * (cc // 2 * 256 + 256 * cc) * 2
        # executed: the 3x3 subnets (Winograd in fp32: / 2.25), the 1x1 subnets as counted
        fwd_ex = 0.0
        for lvl, (mm, cc) in enumerate(((m0, 48), (m0 // 4, 192))):
            for blk in range(args.num_coupling):
                kk = 9 if blk % 2 == 0 else 1
                f = 2.0 * mm * kk * (cc // 2 * 256 + 256 * cc) * 2
                fwd_ex += f / 2.25 if (kk == 9 and not bf16) else f
        roof['step'] = {'alg_tflop_per_step': 6 * fwd / 1e12, 'alg_tflops': 6 * fwd / (ms_per_step * 1e-3) / 1e12,
                        'frac_of_peak_algorithmic': 6 * fwd / (ms_per_step * 1e-3) / 1e12 / peak,
                        'executed_tflops': 6 * fwd_ex / (ms_per_step * 1e-3) / 1e12,
                        'frac_of_peak': 6 * fwd_ex / (ms_per_step * 1e-3) / 1e12 / peak}
    if irn:
        # dominant kernel class of the IRN step = the one with the largest single-stream time; same accounting as above
        # (algorithmic direct-convolution FLOPs / HIP-event bracket on the launch stream, f32 matrix-pipe peak)
        dom = max((c for c in classes if c.get('alg_tflops')), key=lambda c: c['ms_per_step'])
        fwd = 0.0
        for mm, ch in ((m0, 48), (m0 // 4, 192)):
            s1 = min(opt.lr_dims, ch // 2)
            for cin, cout in ((ch - s1, s1), (s1, ch - s1), (s1, ch - s1)):          # F, G, H (archs.py:135-160)
                fwd += args.num_coupling * 2.0 * mm * 9 * (32 * (4 * cin + 192) + (cin + 128) * cout)
        roof = {'bound': 'mfma', 'achieved': dom['executed_tflops'], 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': dom['frac'], 'achieved_algorithmic': dom['alg_tflops'], 'frac_algorithmic': dom['frac_algorithmic'], 'traffic': None,
                'kernel': 'IRN: ' + dom['class'] + ' -- average over all its launches of a step (both levels)',
                'avg_ms': dom['ms_per_step'] / max(dom['launches_per_step'], 1),
                'avg_ms_source': 'HIP events on the launch stream, single-stream steps run right after the timed region',
                'classes': classes, 'single_stream_ms_per_step': iso_ms,
                'step': {'alg_tflop_per_step': 6 * fwd / 1e12, 'alg_tflops': 6 * fwd / (ms_per_step * 1e-3) / 1e12,
                         'frac_of_peak': 6 * fwd / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}}
    out = {'metric': 'training frames/sec at 256x256 bs=16', 'value': value, 'unit': 'frames/s', 'n_gpus': ws,
           'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16' if bf16 else 'f32', 'data': 'synthetic',
           'config': {'workload': (preset['name'] if not custom else f'custom: {args.arch} {args.height}x{args.width}x3, -c {args.num_coupling}, {args.precision}')
                                  + f', synthetic clip ({args.frames} frames, lr_window {args.lr_window}), batch {b}/GPU, '
                                    'full training step (fwd+bwd, rev+bwd, Adam)',
                      'baseline_config': args.config, 'height': args.height, 'width': args.width,
                      'global_batch': ws * b, 'num_coupling': args.num_coupling, 'architecture': args.arch, 'parallelism': f'dp{ws}'},
           'roofline': roof}
    out['distributed'] = distributed_record(ranks, allreduce_ms, ms_per_step)
    if args.with_tcr:
        out['config']['workload'] += f'; + the TCR branch (lambda_bwd_tcr 1, {args.tcr_iters} iteration(s) per step)'
        out['config']['with_tcr'] = args.tcr_iters
    out['gpu_clocks'] = {'before_timed_region': clocks_before, 'after_timed_region': clocks_after,
                         'source': '/sys/class/drm/card*/device/pp_dpm_{sclk,mclk} (active level), gpu_busy_percent'}
    out['config']['hip_graph'] = bool(args.graph == 'on' and any('graph' in v for v in model.__dict__.get('_graphs', {}).values()))
    if args.with_flow:
        out['config']['workload'] += '; preceded in every step by the pair_flow warp + photometric metric + flow gradient on the batch'
        out['config']['with_flow'] = True
    if args.config not in (1, 2) or custom:
        out['metric'] = f'training frames/sec at {args.width}x{args.height} bs={b}' + (' (IRN architecture)' if irn else '')
    if not args.no_cpu_baseline and ws == 1:
        out['cpu_baseline'] = cpu_baseline(args, opt)
    else:
        out['cpu_baseline'] = None
    emit(json.dumps(out))


if __name__ == '__main__':
    main()
