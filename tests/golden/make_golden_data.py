"""Generate golden_data.npz FROM THE REFERENCE'S OWN CODE: the byte / index / flag arithmetic either side of the hot path.

Run once in the build container (needs /root/reference; never runs on the GPU box):
    python tests/golden/make_golden_data.py
The reference modules are imported unmodified.  Third-party packages that are absent here (cv2, colour_demosaicing,
imageio, pytorch_lightning, torchvision, kornia, FrEIA) are registered as EMPTY namespace modules that only carry the names
the import lines and class statements need -- the trick make_golden.py uses for FrEIA.  None of their arithmetic is
emulated; where the reference hands data to one of them the stub RECORDS the operands (that is how the packed Bayer mosaic
of prepare.py:103-116 and the rotation inputs / translation column of tcr.py:26-45 are captured) and nothing downstream of a
stub is stored.  Outputs are data only.

  G8  datasets/prepare.py: extract_bayer (:35-52, scale 1: cv2.resize with fx = fy = 1 is the identity, the stub returns its
      input), binning (:54-82, mean / sum, scales 1 / 2 / 4), the uint8 quantisation of :127-128 / :164, and the mosaic that
      pack_demosaic (:103-116) hands to the demosaicer
  G9  data.py: populate_files of the three datasets (:55-59, :72-76, :87-99) on real temp directory listings (num_lr =
      len(listdir) - 1, :22), ConcatDataset.__len__ / the unsupervised index stream of __getitem__ (:105-118)
  G10 main.py get_args (:9-83): derived lr_dims / z_dims (:74-75), defaults and types of every flag
  G11 tcr.py TCR.forward (:26-45): centre, angle and zoom handed to kornia.get_rotation_matrix2d and the translation added to
      the matrix (incl. the division by `scale`)
"""
import os, sys, tempfile, types
import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
RECORD = {}


def _ns(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def register_namespace_stubs():
    class _Base:                       # a class statement needs a base to exist; no behaviour
        def __init__(self, *a, **k):
            pass

    def record_demosaic(cfa, *a, **k):
        RECORD['cfa'] = np.array(cfa, copy=True)
        return np.zeros(cfa.shape + (3,))

    def record_rotation(center, angle, zoom):
        RECORD['center'], RECORD['angle'], RECORD['zoom'] = center.clone(), angle.clone(), zoom.clone()
        return torch.zeros(center.shape[0], 2, 3)

    def record_warp(img, mat, dsize=None, **k):
        RECORD['mat'], RECORD['dsize'] = mat.clone(), tuple(dsize)
        return img

    def identity_resize(frame, dsize, fx=None, fy=None, interpolation=None):
        assert fx == 1 and fy == 1, 'only the scale-1 call (a same-size resize) is recorded'
        return frame

    _ns('cv2', resize=identity_resize, INTER_LANCZOS4=4)
    _ns('colour_demosaicing', demosaicing_CFA_Bayer_bilinear=record_demosaic)
    _ns('imageio')
    _ns('kornia', get_rotation_matrix2d=record_rotation, warp_affine=record_warp)
    for name in ('FrEIA', 'FrEIA.framework', 'FrEIA.modules', 'torchvision'):
        _ns(name)
    sys.modules['torchvision'].transforms = _ns('torchvision.transforms')
    pl = _ns('pytorch_lightning', LightningModule=torch.nn.Module, LightningDataModule=_Base, Trainer=_Base)
    pl.loggers = _ns('pytorch_lightning.loggers', WandbLogger=_Base)
    pl.callbacks = _ns('pytorch_lightning.callbacks', ModelCheckpoint=_Base)
    pl.callbacks.progress = _ns('pytorch_lightning.callbacks.progress', ProgressBarBase=_Base)


def import_reference(name, path):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.path.insert(0, REF)
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.path.pop(0)
    return mod


def main():
    register_namespace_stubs()
    out = {}

    # ---- G8 datasets/prepare.py ------------------------------------------------------------------------------------
    prepare = import_reference('ref_prepare', os.path.join(REF, 'datasets', 'prepare.py'))
    rng = np.random.RandomState(8)
    frames = rng.randint(0, 256, (3, 32, 48, 3)).astype(np.uint8)
    frames[1, :8] = 255                           # saturated rows: 'sum' clips, 'mean' does not
    frames[2, :, :16] = rng.randint(0, 4, (32, 16, 3))   # near-black: floor() of small means
    out['g8_frames'] = frames
    for t, frame_u8 in enumerate(frames):
        frame = frame_u8 / 255                                            # prepare.py:127-128
        bayer, hr = prepare.extract_bayer(frame, 1)                       # :35-52
        out[f'g8_bayer_{t}'] = bayer
        assert np.array_equal((np.clip(hr, 0, 1) * 255).astype(np.uint8), frame_u8)   # :139-140 round trip of u8 frames
        for reduction in ('mean', 'sum'):
            for scale in (1, 2, 4):
                lr = prepare.binning(bayer, reduction, scale)             # :54-82
                out[f'g8_binned_{t}_{reduction}_{scale}'] = lr
                out[f'g8_lr_u8_{t}_{reduction}_{scale}'] = (np.clip(lr, 0, 1) * 255).astype(np.uint8)   # :164
                prepare.pack_demosaic(lr)                                 # :103-116 (the demosaicer records its operand)
                out[f'g8_cfa_{t}_{reduction}_{scale}'] = RECORD.pop('cfa')

    # ---- G9 data.py ------------------------------------------------------------------------------------------------
    ref_data = import_reference('data', os.path.join(REF, 'data.py'))
    cases = [(241, 10, 10, 7), (301, 10, 2, 7), (8, 1, 1, 3), (200, 30, 3, 12), (64, 5, 1, 0), (400, 120, 0, 9)]
    out['g9_cases'] = np.array(cases, dtype=np.int64)

    def numbers(paths):
        return np.array([int(os.path.basename(p)[6:11]) for p in paths], dtype=np.int64)

    with tempfile.TemporaryDirectory() as root:
        for ci, (listed, fps, win, k) in enumerate(cases):
            scene = f'scene{ci}'
            for sub in ('lr_frames', 'hr_frames'):
                os.makedirs(os.path.join(root, sub, scene))
            for t in range(listed):                      # `listed` directory entries -> num_lr = listed - 1 (data.py:22)
                open(os.path.join(root, 'lr_frames', scene, f'frame_{t:05d}.png'), 'wb').close()
            opt = types.SimpleNamespace(dataset=root, scene=scene, fps=fps, lr_window=win, operation='train')
            sup, unsup = ref_data.VideoTrainDataset(opt), ref_data.VideoAllDataset(opt)
            out[f'g9_{ci}_train'], out[f'g9_{ci}_all'] = numbers(sup.hr_files), numbers(unsup.hr_files)
            out[f'g9_{ci}_train_window0'] = numbers(sup.lr_files[0]) if len(sup) else np.zeros(0, np.int64)
            assert all(os.path.dirname(p).endswith(os.path.join('hr_frames', scene)) for p in sup.hr_files)
            torch.manual_seed(100 + ci)
            val = ref_data.VideoValDataset(opt, k)
            out[f'g9_{ci}_val'] = numbers(val.hr_files)
            out[f'g9_{ci}_val_window0'] = numbers(val.lr_files[0]) if len(val) else np.zeros(0, np.int64)
            out[f'g9_{ci}_shuffle'] = np.array([sup.shuffle, unsup.shuffle, val.shuffle])
            cd = ref_data.ConcatDataset(list(range(len(sup))), list(range(len(unsup))))
            torch.manual_seed(200 + ci)
            out[f'g9_{ci}_pairs'] = np.array([cd[i] for i in range(len(cd))], dtype=np.int64).reshape(-1, 2)
            out[f'g9_{ci}_len'] = np.int64(len(cd))
            opt.operation = 'test'
            out[f'g9_{ci}_all_shuffle_test'] = np.bool_(ref_data.VideoAllDataset(opt).shuffle)

    # ---- G10 main.py get_args --------------------------------------------------------------------------------------
    sys.modules['data'] = ref_data
    _ns('lit_wrapper', SingleVideoINN=object)             # main.py:7 only needs the name; lit_wrapper is not exercised
    ref_main = import_reference('ref_main', os.path.join(REF, 'main.py'))
    argvs = [['train'], ['train', '--lr_window', '1', '--fps', '1'], ['train', '--lr_window', '0'],
             ['train', '--lr_window', '23', '-c', '12', '-a', 'IRN', '-b', '16'], ['train', '--scale', '8', '--lr_window', '2']]
    for ai, argv in enumerate(argvs):
        saved = sys.argv
        sys.argv = ['main.py'] + argv
        try:
            args = ref_main.get_args()
        finally:
            sys.argv = saved
        out[f'g10_{ai}_argv'] = np.array(argv)
        out[f'g10_{ai}_dims'] = np.array([args.lr_dims, args.z_dims], dtype=np.int64)
        if ai == 0:
            d = vars(args)
            out['g10_default_names'] = np.array(sorted(d))
            out['g10_default_values'] = np.array([repr(d[k]) for k in sorted(d)])
            out['g10_default_types'] = np.array([type(d[k]).__name__ for k in sorted(d)])

    # ---- G11 tcr.py ------------------------------------------------------------------------------------------------
    ref_tcr = import_reference('ref_tcr', os.path.join(REF, 'tcr.py'))
    g = torch.Generator().manual_seed(11)
    for ti, (shape, ang, trans, scale) in enumerate([((3, 3, 32, 48), 5.0, 5.0, 1), ((4, 84, 8, 6), 5.0, 5.0, 0.25),
                                                    ((2, 3, 17, 9), 30.0, 2.5, 1)]):
        rand = torch.rand(shape[0], 3, generator=g)
        ref_tcr.TCR(ang, trans)(torch.zeros(shape), rand, scale)
        out.update({f'g11_{ti}_cfg': np.array(list(shape) + [ang, trans, scale], dtype=np.float64), f'g11_{ti}_rand': rand,
                    f'g11_{ti}_center': RECORD['center'], f'g11_{ti}_angle': RECORD['angle'], f'g11_{ti}_zoom': RECORD['zoom'],
                    f'g11_{ti}_translation': RECORD['mat'][:, :, 2], f'g11_{ti}_dsize': np.array(RECORD['dsize'])})

    out = {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}
    path = os.path.join(HERE, 'golden_data.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes,', len(out), 'arrays')


if __name__ == '__main__':
    main()
