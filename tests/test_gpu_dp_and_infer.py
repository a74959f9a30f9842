"""GPU: the data-parallel path (2 ranks, one flat-gradient all-reduce per step) against a single process that sees
the concatenated batch, and the inference surface (SingleVideoINN.infer)."""
import os
import subprocess
import sys
import types

import pytest
import torch

from conftest import free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_DP_SCRIPT = r'''
import os, sys, types, torch
sys.path.insert(0, %(root)r)
import sin_inn_amd
from sin_inn_amd import dist as sd
# two ranks share the single GPU of the test box -> gloo (RCCL refuses two ranks on one device); the code path
# (flat gradient buffer, one all-reduce before the fused Adam launch, rank-0 weight broadcast) is the same
rank, ws = sd.init_from_env('gloo')
torch.cuda.set_device(0)
import lit_wrapper
from data import FrameStore
from sin_inn_amd.functional import sample_windows
sys.path.insert(0, os.path.join(%(root)r, 'tests'))
from test_gpu_model import make_opt
opt = make_opt(num_coupling=1)
torch.manual_seed(123)
model = lit_wrapper.SingleVideoINN(3, 32, 32, opt).cuda()
optim = model.attach_optimizer()
sd.broadcast_([p.data for p in model.parameters()])
store = FrameStore.synthetic(12, 32, 32)
g = torch.Generator().manual_seed(5)
zs = torch.randn(4, opt.z_dims, 4, 4, generator=g)
idx_all = torch.tensor([2, 3, 5, 7])
mine = idx_all[rank::ws] if ws > 1 else idx_all
zmine = zs[rank::ws] if ws > 1 else zs
lit_wrapper._latent = lambda b, zd, h, w, device, temp=1.0: zmine.to(device)
import time
steps, delay = int(os.environ.get('DP_STEPS', '2')), float(os.environ.get('DP_DELAY', '0'))
for step in range(steps):
    time.sleep(delay * (rank if step %% 2 == 0 else ws - 1 - rank))     # unequal host delays: ranks reach the collective apart
    hr, lr = sample_windows(store.hr.cuda(), store.lr.cuda(), mine.cuda(), 1)
    model.training_step([{'hr': hr, 'lr': lr}, {'hr': hr, 'lr': lr}], 0)
flat = optim.flat_params()[0].detach().cpu()
torch.save(flat, os.path.join(%(out)r, f'params_ws{ws}_rank{rank}_s{steps}.pt'))
print('rank', rank, 'of', ws, 'done')
'''


def _run(tmp_path, nproc, port, steps=2, delay=0.0):
    script = tmp_path / f'dp{nproc}.py'
    script.write_text(_DP_SCRIPT % dict(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0', DP_STEPS=str(steps), DP_DELAY=str(delay))
    if nproc == 1:
        cmd = [sys.executable, str(script)]
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
               '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]


def test_two_rank_data_parallel_equals_single_process_on_the_global_batch(tmp_path):
    _run(tmp_path, 2, free_port())
    _run(tmp_path, 1, 0)
    r0 = torch.load(tmp_path / 'params_ws2_rank0_s2.pt')
    r1 = torch.load(tmp_path / 'params_ws2_rank1_s2.pt')
    single = torch.load(tmp_path / 'params_ws1_rank0_s2.pt')
    assert torch.equal(r0, r1)                                   # replicas stay bit-identical
    # mean of per-rank mean-losses == global-batch mean -> same update as one process on all 4 samples
    err = float((r0 - single).abs().max())
    assert err <= 2.5e-4, err                                    # 2 Adam steps of lr 1e-4 (sign-like on noise-level grads)
    big = (single - single.mean()).abs() > 0
    assert float((r0 - single).abs().mean()) < 2e-6


def test_replicas_stay_bit_identical_under_unequal_host_delays(tmp_path):
    """5 steps, the ranks delayed against each other before every step (alternating which one is late): the gradient
    all-reduce is ordered behind the weight-gradient stream and Adam behind the all-reduce by stream waits alone, so however
    far apart the hosts are the replicas' weights stay bitwise equal."""
    _run(tmp_path, 2, free_port(), steps=5, delay=0.05)
    r0 = torch.load(tmp_path / 'params_ws2_rank0_s5.pt')
    r1 = torch.load(tmp_path / 'params_ws2_rank1_s5.pt')
    assert torch.isfinite(r0).all() and torch.equal(r0, r1)


_ORDER_SCRIPT = r'''
import os, sys, torch
sys.path.insert(0, %(root)r)
import sin_inn_amd
from sin_inn_amd import dist as sd
import torch.distributed as dist
rank, ws = sd.init_from_env('nccl', allow_single=True)          # a ONE-rank RCCL group: the real backend, the real streams
assert dist.is_initialized() and dist.get_backend() == 'nccl' and ws == 1
side = torch.cuda.Stream()
buf = torch.ones(1 << 22, device='cuda')
torch.cuda.synchronize()
with torch.cuda.stream(side):                                   # the producer of the buffer, held up on the side stream
    torch.cuda._sleep(200_000_000)
    buf.mul_(3.0)
sd.allreduce_sum_([buf], after=side, _force=True)               # issued from the main stream, ordered behind `side` alone
out = buf * 2.0                                                 # main stream: must see the reduced (== produced) values
torch.cuda.synchronize()
assert float(out.min()) == 6.0 and float(out.max()) == 6.0, (float(out.min()), float(out.max()))
sys.stdout.write('ordering ok\n')
dist.destroy_process_group()
'''


def test_gradient_allreduce_is_ordered_behind_the_producer_stream_on_rccl(tmp_path):
    """dist.allreduce_sum_(after=stream) on the REAL backend (RCCL, a one-rank group -- two ranks cannot share the test box's
    GPU): the collective waits for a producer that is still spinning on the side stream although it is issued from the main
    stream, and the main stream's next kernel waits for the collective.  Exercises ProcessGroupNCCL.Options(high priority),
    async_op under a stream context and work.wait() exactly as the optimiser proxy uses them."""
    script = tmp_path / 'order.py'
    script.write_text(_ORDER_SCRIPT % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and 'ordering ok' in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_infer_writes_frames(tmp_path):
    import lit_wrapper
    from data import FrameStore, VideoAllDataset, get_loader
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_gpu_model import make_opt
    opt = make_opt(num_coupling=1)
    opt.frame_store = FrameStore.synthetic(10, 32, 32)
    opt.operation = 'test'
    torch.manual_seed(0)
    model = lit_wrapper.SingleVideoINN(3, 32, 32, opt).cuda()
    data = VideoAllDataset(opt)
    outs = model.infer(get_loader(data, 40), opt)
    assert len(outs) == 1 and outs[0].shape == (len(data), 3, 32, 32) and torch.isfinite(outs[0]).all()
    model.infer(get_loader(data, 40), opt, save_images=str(tmp_path / 'frames'))
    assert len(os.listdir(tmp_path / 'frames')) == len(data)


def test_cli_training_under_two_ranks(tmp_path):
    """`main.py train` launched with torch.distributed.run, 2 ranks (gloo: they share the test box's single GPU): rank shards of
    unequal natural length (3 training frames over 2 ranks) still run the same number of steps, rank 0 writes the checkpoint."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0', SININN_DIST_BACKEND='gloo',
               SININN_FORCE_DEVICE='0')
    wd = str(tmp_path / 'exp')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'main.py'), 'train', '--synthetic', '52', '32', '32', '--fps', '10',
           '--lr_window', '1', '-c', '1', '-b', '1', '-e', '2', '--save_iter', '2', '-p', '1', '-w', wd, '--suffix', 'dp']
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    import glob
    ckpts = glob.glob(os.path.join(wd, 'train', '*', 'checkpoints', 'epoch=1.ckpt'))
    assert len(ckpts) == 1
    ck = torch.load(ckpts[0], map_location='cpu')
    # 52 frames, fps 10 -> train frames [11, 23, 35]: 3 positions over 2 ranks -> padded to 4 -> 2 steps per rank per epoch
    assert ck['global_step'] == 4
