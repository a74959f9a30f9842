"""GPU: the drop-in surface end to end -- CLI train -> checkpoint -> resume -> test (reference main.py:85-137), validation_step
(lit_wrapper.py:79-89), inference frames against the oracle inverse (lit_wrapper.py:91-128), the PNG directory layout
(data.py:14-45), and the packed-weight hand-over between the two pass streams."""
import glob
import json
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def _cli(*extra):
    return ['--synthetic', '40', '32', '32', '--fps', '10', '--lr_window', '1', '-c', '1', '-b', '2', '--suffix', 'e2e'] \
        + list(extra)


def test_cli_train_checkpoint_resume_test(tmp_path):
    """main.py train (2 epochs, checkpoint each) -> main.py train --resume_state (continues at epoch 2) -> main.py test
    writing one frame per LR window.  40 synthetic frames, fps 10 -> train frames [11, 23] (data.py:56), 18 test frames."""
    import main
    from sin_inn_amd.lightning import load_checkpoint
    wd = str(tmp_path / 'exp')
    main.main(['train'] + _cli('-e', '2', '--save_iter', '1', '-p', '1', '-w', wd))
    ckpts = sorted(glob.glob(os.path.join(wd, 'train', '*', 'checkpoints', 'epoch=*.ckpt')))
    assert [os.path.basename(c) for c in ckpts] == ['epoch=0.ckpt', 'epoch=1.ckpt']
    assert os.path.getsize(ckpts[1]) < 40e6                                   # weights + Adam state, not the clip
    ck1 = load_checkpoint(ckpts[1], map_location='cpu')
    assert ck1['epoch'] == 1 and ck1['global_step'] == 2 and 'frame_store' not in ck1['hyper_parameters']['opt']
    log = [json.loads(l) for l in open(glob.glob(os.path.join(wd, 'train', '*', '*.jsonl'))[0])]
    assert any('lr_acc' in r and 'hr_acc' in r and 'z_nll' in r for r in log)   # validation ran and was logged
    # resume: one more epoch; weights move on from the checkpoint, Adam state is carried (step counter 2 -> 3)
    main.main(['train'] + _cli('-e', '3', '--save_iter', '1', '-p', '5', '-w', wd, '-r', ckpts[1]))
    ck2 = load_checkpoint(os.path.join(os.path.dirname(ckpts[1]), 'epoch=2.ckpt'), map_location='cpu')
    assert ck2['epoch'] == 2 and ck2['global_step'] == 3
    assert ck2['optimizer_states'][0]['flat'][0]['step'] == 3
    k = 'inn.module_list.3.s1.0.weight'
    assert not torch.equal(ck1['state_dict'][k], ck2['state_dict'][k])
    assert float((ck1['state_dict'][k] - ck2['state_dict'][k]).abs().max()) < 1e-3      # one Adam step of lr 1e-4
    # test: strict load of the checkpoint, inverse pass over every window, PNG frames
    frames = str(tmp_path / 'frames')
    model = main.main(['test'] + _cli('-w', wd, '-r', ckpts[1], '--save_images', frames))
    assert len(os.listdir(frames)) == 18
    assert torch.equal(model.state_dict()[k].cpu(), ck1['state_dict'][k])
    # a checkpoint of another architecture must not be loaded silently
    other = str(tmp_path / 'other.ckpt')
    torch.save({'state_dict': {'inn.nothing': torch.zeros(1)}}, other)
    with pytest.raises(SystemExit):
        main.main(['test'] + _cli('-w', wd, '-r', other, '--save_images', frames))


def test_baseline_config0_through_the_cli_matches_the_oracle_step(tmp_path):
    """BASELINE configs[0] exactly as named: the 4-coupling-block SRF network on 8 synthetic 64x64x3 frames through main.py
    (`--synthetic 8 64 64 --fps 1 --lr_window 1 -c 4`: num_lr 7, ONE supervised frame [2] (data.py:56), lr_dims 12, z_dims 180).  One
    epoch = one training step on the GPU; the CPU oracle runs the same step from the same seed-0 weights, frame window and latent:
    logged loss within 1e-4, and the Adam update agrees element by element (an update is lr * sign(g) for all but vanishing
    gradients: entries that differ by more than a tenth of the step are counted, not bounded)."""
    import lit_wrapper
    import main
    from data import ConcatDataset, VideoAllDataset, VideoTrainDataset, VideoValDataset
    from oracle import sininn_oracle as O
    argv = ['train', '--synthetic', '8', '64', '64', '--fps', '1', '--lr_window', '1', '-c', '4', '-b', '8', '-e', '1', '--save_iter', '100',
            '-p', '100', '-w', str(tmp_path / 'exp'), '--suffix', 'cfg0']
    z = torch.randn(1, 180, 8, 8, generator=torch.Generator().manual_seed(2))
    real = lit_wrapper._latent
    lit_wrapper._latent = lambda b, zd, h, w, device, temp=1.0: z[:b].to(device)
    try:
        model = main.main(argv)
    finally:
        lit_wrapper._latent = real
    assert model.opt.lr_dims == 12 and model.opt.z_dims == 180
    # the same construction sequence main() runs (it consumes the seeded RNG in this order), stopped before training
    args = main.get_args(argv)
    sup, unsup = VideoTrainDataset(args), VideoAllDataset(args)
    train = ConcatDataset(sup, unsup)
    VideoValDataset(args, len(train) * 4 // 6)
    assert sup.frames == [2] and len(train) == 1
    fresh = lit_wrapper.SingleVideoINN(3, 64, 64, args)
    ref = O.SRFlowOracle(3, 64, 64, scale=4, num_coupling=4)
    ref.load_state_dict({k[len('inn.'):]: v.detach().clone() for k, v in fresh.state_dict().items()})
    before = torch.cat([p.detach().reshape(-1) for p in ref.parameters()]).clone()
    hr, lr = O.gather_window(args.frame_store.lr, args.frame_store.hr, 2, 1)
    adam = torch.optim.Adam(ref.parameters(), lr=args.learning_rate, betas=tuple(args.adam_betas), weight_decay=args.weight_decay)
    lam = dict(fwd_rec=1.0, fwd_mmd=0.0, latent_nll=0.0, bwd_rec=1.0, bwd_mmd=0.0)
    fwd, bwd, _, _, _ = O.training_step(ref, hr[None], lr[None], z, lam, 12)
    adam.step()
    assert abs(float(model._logged['train']) / float(fwd + bwd) - 1) < 1e-4
    want = torch.cat([p.detach().reshape(-1) for p in ref.parameters()]) - before
    got = torch.cat([p.detach().reshape(-1).cpu() for p in model.inn.parameters()])[:before.numel()] - before
    assert float(want.abs().max()) > 0.5 * args.learning_rate                 # a step was taken
    off = (got - want).abs() > 0.1 * args.learning_rate
    assert float(off.float().mean()) < 2e-3, float(off.float().mean())


def _model_and_oracle(size=32, num_coupling=1, lr_window=1, seed=0, **kw):
    import lit_wrapper
    from oracle import sininn_oracle as O
    from test_gpu_model import make_opt
    torch.manual_seed(seed)
    opt = make_opt(num_coupling=num_coupling, lr_window=lr_window, **kw)
    model = lit_wrapper.SingleVideoINN(3, size, size, opt)
    ref = O.SRFlowOracle(3, size, size, scale=4, num_coupling=num_coupling)
    ref.load_state_dict({k[len('inn.'):]: v.detach().clone() for k, v in model.state_dict().items()})
    return model.cuda(), ref, opt


def test_validation_step_values_match_oracle():
    """lit_wrapper.py:79-89: lr_acc / hr_acc / z_nll of one forward + one inverse pass under no_grad."""
    import lit_wrapper
    from oracle import sininn_oracle as O
    model, ref, opt = _model_and_oracle(size=64, num_coupling=2)
    g = torch.Generator().manual_seed(4)
    hr = torch.rand(3, 3, 64, 64, generator=g)
    lr = torch.rand(3, opt.lr_dims, 8, 8, generator=g)
    z = torch.randn(3, opt.z_dims, 8, 8, generator=g)
    real = lit_wrapper._latent
    lit_wrapper._latent = lambda b, zd, h, w, device, temp=1.0: z.to(device)
    try:
        with torch.no_grad():
            model.validation_step({'hr': hr.cuda(), 'lr': lr.cuda()}, 0)
    finally:
        lit_wrapper._latent = real
    with torch.no_grad():
        lr_z_hat = ref(hr)
        hr_hat = ref(torch.cat((lr, z), 1), rev=True)
    want = dict(lr_acc=O.reconstruction(lr_z_hat[:, :opt.lr_dims], lr), hr_acc=O.reconstruction(hr_hat, hr),
                z_nll=O.latent_nll(lr_z_hat[:, opt.lr_dims:]))
    for k, v in want.items():
        assert abs(float(model._logged[k]) / float(v) - 1) < 1e-4, k


@pytest.mark.parametrize('mode', ['clamp', 'wrap'])
def test_infer_frames_match_oracle_inverse(tmp_path, mode):
    """The PNGs infer() writes are the oracle's inverse pass, converted like the reference converts (mode 'wrap' =
    ToPILImage's mul(255).byte()) or clamped first (mode 'clamp', the default); +-1 LSB for the 1e-4 float tolerance."""
    import lit_wrapper
    from PIL import Image
    from data import FrameStore, VideoAllDataset, get_loader
    model, ref, opt = _model_and_oracle(size=32, num_coupling=2)
    # a random-init network fed z ~ N(0, temp^2) leaves [0,1] in many pixels, so clamp and wrap really differ here
    opt.frame_store = FrameStore.synthetic(40, 32, 32)
    opt.operation, opt.pixel_mode, opt.temp = 'test', mode, 0.8
    data = VideoAllDataset(opt)
    n = len(data)
    z = torch.randn(n, opt.z_dims, 4, 4, generator=torch.Generator().manual_seed(8))
    real = lit_wrapper._latent
    lit_wrapper._latent = lambda b, zd, h, w, device, temp=1.0: (z * temp).to(device)
    try:
        model.infer(get_loader(data, 40), opt, save_images=str(tmp_path / 'f'))
    finally:
        lit_wrapper._latent = real
    lr = data.batch(range(n))['lr'].cpu()
    with torch.no_grad():
        hr_hat = ref(torch.cat((lr, z * opt.temp), 1), rev=True)
    if mode == 'wrap':
        want = hr_hat.mul(255).to(torch.int32).remainder(256).to(torch.uint8)     # == .byte() of ToPILImage
    else:
        want = (hr_hat.clamp(0, 1) * 255).to(torch.uint8)
    assert float(((hr_hat < 0) | (hr_hat > 1)).float().mean()) > 0.01              # the case the modes differ on exists
    files = sorted(os.listdir(tmp_path / 'f'))
    assert len(files) == n and files[0] == 'out_0000_00.png'
    got = torch.stack([torch.from_numpy(np.asarray(Image.open(tmp_path / 'f' / f))) for f in files]).permute(0, 3, 1, 2)
    diff = (got.int() - want.int()).abs()
    diff = torch.minimum(diff, 256 - diff) if mode == 'wrap' else diff             # 255 <-> 0 is one step when wrapping
    # float tolerance of the path (1e-4 of the max-norm) in LSBs: a pixel may differ by one step only where the oracle
    # value sits within that distance of a truncation boundary (an integer of 255*x; 0 and 1 themselves when clamping)
    tol = 255.0 * 1e-4 * float(hr_hat.abs().max()) + 1e-3
    scaled = hr_hat * 255
    near_edge = (scaled - scaled.round()).abs() < tol
    assert int(diff.max()) <= 1 and not bool(((diff > 0) & ~near_edge).any()), (int(diff.max()), tol)


def test_frames_to_u8_modes_bit_exact():
    from sin_inn_amd.functional import frames_to_u8
    g = torch.Generator().manual_seed(1)
    x = torch.rand(3, 3, 17, 23, generator=g) * 3 - 1                                # [-1, 2)
    for layout in ('nchw', 'nhwc'):
        xg = x.cuda() if layout == 'nchw' else x.cuda().contiguous(memory_format=torch.channels_last)
        got = frames_to_u8(xg, wrap=False).cpu()
        assert torch.equal(got, (x.clamp(0, 1) * 255).to(torch.uint8).permute(0, 2, 3, 1))
        got = frames_to_u8(xg, wrap=True).cpu()
        assert torch.equal(got, x.mul(255).to(torch.int32).remainder(256).to(torch.uint8).permute(0, 2, 3, 1))


def test_png_directory_to_training_batch(tmp_path):
    """FrameStore.from_directory (data.py:14-45 layout) -> sampler kernel == the oracle's gather on the same files."""
    from oracle import sininn_oracle as O
    from test_host_cpu import _write_png_tree
    import data
    hr, lr = _write_png_tree(str(tmp_path), 'clip', 30, 32, 32)
    opt = types.SimpleNamespace(fps=10, lr_window=2, operation='train', dataset=str(tmp_path), scene='clip')
    ds = data.VideoAllDataset(opt)
    assert ds.frames == list(range(11, 19))
    batch = next(iter(data.get_loader(ds, 4)))
    idx = None
    # shuffle=True for the 'train' operation: recover which frames were drawn from the batch itself
    for n in range(4):
        match = [t for t in ds.frames if torch.equal(batch['hr'][n].cpu(), O.gather_window(torch.from_numpy(lr), torch.from_numpy(hr), t, 2)[0])]
        assert len(match) == 1
        assert torch.equal(batch['lr'][n].cpu(), O.gather_window(torch.from_numpy(lr), torch.from_numpy(hr), match[0], 2)[1])


def test_pack_miss_after_validation_is_ordered_across_streams():
    """packs rebuilt on a cache miss (first step; first training step after a no_grad validation; weights
    changed through torch) must be complete before the OTHER pass chain reads them.  GPU-bound size so the second stream
    really runs behind; the overlapped run must equal the single-stream run bit for bit."""
    import lit_wrapper
    from data import FrameStore
    from sin_inn_amd.functional import sample_windows
    from test_gpu_model import make_opt

    def run(overlap):
        torch.manual_seed(3)
        opt = make_opt(num_coupling=2, lr_window=2)
        model = lit_wrapper.SingleVideoINN(3, 256, 256, opt).cuda()
        model.overlap_passes = overlap
        optim = model.attach_optimizer()
        store = FrameStore.synthetic(12, 256, 256).to('cuda')
        g = torch.Generator().manual_seed(7)
        torch.cuda.manual_seed(9)
        for it in range(3):
            idx = torch.randint(2, 10, (8,), generator=g).cuda()
            hr, lr = sample_windows(store.hr, store.lr, idx, 2)
            if it == 1:
                with torch.no_grad():
                    model.validation_step({'hr': hr, 'lr': lr}, 0)
                for p in model.parameters():                 # weights touched through torch: every pack is stale now
                    p.data.mul_(1.0009765625)
            model.training_step([{'hr': hr, 'lr': lr}, {'hr': hr, 'lr': lr}], 0)
        torch.cuda.synchronize()
        return optim.flat_params()[0].clone()

    a, b = run(True), run(False)
    assert torch.isfinite(a).all() and torch.equal(a, b)


def test_bench_with_flow_line_is_well_formed(tmp_path):
    """`bench.py --with-flow` (BASELINE configs[3] as named: pair_flow warp + INN) at a toy size: one JSON line with the contract's
    keys, the roofline / cpu_baseline objects and the two HBM rows of the warp."""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', '3', '--with-flow', '--size', '64', '--batch', '2',
                          '--num-coupling', '1', '--frames', '24', '--steps', '2', '--warmup', '1', '--cpu-batch', '1'],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in rec, k
    assert rec['dtype'] == 'bf16' and rec['config']['with_flow'] is True and rec['value'] > 0
    r = rec['roofline']
    assert r['bound'] in ('hbm', 'mfma') and 0 < r['frac'] <= 1.0 and r['achieved'] > 0 and r['peak'] > 0
    warp = [c for c in r['classes'] if c['class'].startswith('flow warp')]
    assert len(warp) == 2 and all(c['bound'] == 'hbm' and c['achieved_gbs'] > 0 for c in warp)
    assert rec['cpu_baseline']['kind'] == 'port' and rec['cpu_baseline']['cores'] >= 1
