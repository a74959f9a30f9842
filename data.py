"""Single-video datasets + frame-window sampler (drop-in for the reference's data.py), MI355X build.

The reference decodes ``2*lr_window+1`` LR PNGs + one HR PNG per sample in 4 DataLoader worker processes on
every step (data.py:31-45,122,134).  A single video is small next to 288 GB of HBM, so here the whole clip is
decoded ONCE, kept resident on the GPU as uint8 (``FrameStore``), and a batch is produced by one HIP gather kernel
(u8 -> f32/255, window -> channels; sin-inn_amd functional.sample_windows).  The index arithmetic of the three
dataset flavours, the random supervised/unsupervised pairing and the loader/batch structure are the reference's:
  train : range(1+fps, num_lr-fps, 120//fps)                  (data.py:55-59)
  all   : range(1+fps, num_lr-fps)                            (data.py:72-76)
  val   : randperm(num_lr-2*win)+win, skipping train frames   (data.py:87-99)
  pair  : (sup[i], unsup[randint(num_unsup)])                 (data.py:112-115)
"""
import os

import numpy as np
import torch

import sin_inn_amd.lightning as pl
from sin_inn_amd import dist as sdist
from sin_inn_amd.functional import sample_windows


class FrameStore:
    """All frames of one video as uint8 tensors: hr (T,H,W,3), lr (T,h,w,4)."""

    def __init__(self, hr_u8, lr_u8, num_listed=None):
        assert hr_u8.dtype == torch.uint8 and lr_u8.dtype == torch.uint8
        assert hr_u8.shape[0] == lr_u8.shape[0] and hr_u8.shape[-1] == 3 and lr_u8.shape[-1] == 4
        self.hr, self.lr = hr_u8.contiguous(), lr_u8.contiguous()
        # the reference counts directory entries minus one (data.py:22)
        self.num_lr = (num_listed if num_listed is not None else hr_u8.shape[0]) - 1

    def to(self, device):
        self.hr, self.lr = self.hr.to(device), self.lr.to(device)
        return self

    @property
    def device(self):
        return self.hr.device

    @classmethod
    def synthetic(cls, frames, height, width, seed_hr=0, seed_lr=1, scale=8):
        """i.i.d. uniform uint8 frames, HR seed 0, LR seed 1."""
        g = torch.Generator().manual_seed(seed_hr)
        hr = torch.randint(0, 256, (frames, height, width, 3), generator=g, dtype=torch.uint8)
        g = torch.Generator().manual_seed(seed_lr)
        lr = torch.randint(0, 256, (frames, height // scale, width // scale, 4), generator=g, dtype=torch.uint8)
        return cls(hr, lr)

    @classmethod
    def from_hr_clip(cls, hr_u8, scale=4, reduction='mean'):
        """Raw RGB video (T,H,W,3) u8 on the GPU -> frame store, LR frames synthesised on-device exactly as
        datasets/prepare.py does offline (RGGB sampling + binning; HR/LR ratio = 2*scale), no PNG round trip."""
        from sin_inn_amd.functional import bayer_bin
        return cls(hr_u8, bayer_bin(hr_u8.contiguous(), scale, reduction))

    @classmethod
    def from_directory(cls, dataset, scene):
        """``<dataset>/{hr_frames,lr_frames}/<scene>/frame_%05d.png`` (data.py:20-21,57-59); LR PNGs are RGBA-coded RGGB."""
        from PIL import Image
        lr_dir = os.path.join(dataset, 'lr_frames', scene)
        hr_dir = os.path.join(dataset, 'hr_frames', scene)
        listed = len(os.listdir(lr_dir))
        names = sorted(f for f in os.listdir(lr_dir) if f.startswith('frame_') and f.endswith('.png'))
        if not names:
            raise FileNotFoundError(f'no frame_%05d.png files in {lr_dir}')
        count = int(names[-1][6:11]) + 1
        # the reference opens files by index (io.imread of frame_{x:05d}.png, data.py:33-38) and so fails on the first gap;
        # a resident store must not paper over one with zero frames either
        gaps = sorted(set(range(count)) - {int(n[6:11]) for n in names})          # the reference indexes from frame_00000.png
        if gaps:
            raise FileNotFoundError(f'{lr_dir}: LR frame(s) missing: ' + ', '.join(f'frame_{t:05d}.png' for t in gaps[:8]))
        hr, lr = None, None
        have_hr = np.zeros(count, bool)
        for name in names:
            t = int(name[6:11])
            a = np.asarray(Image.open(os.path.join(lr_dir, name)))
            if lr is None:
                lr = np.zeros((count,) + a.shape, np.uint8)
            lr[t] = a
            hp = os.path.join(hr_dir, name)
            if os.path.isfile(hp):
                b = np.asarray(Image.open(hp))[..., :3]
                if hr is None:
                    hr = np.zeros((count,) + b.shape, np.uint8)
                hr[t] = b
                have_hr[t] = True
        if hr is None:
            raise FileNotFoundError(f'no HR frames for scene {scene!r} in {hr_dir}')
        store = cls(torch.from_numpy(hr), torch.from_numpy(lr), num_listed=listed)
        store.have_hr = have_hr
        return store

    def require_hr(self, frames):
        """Raise (like the reference's io.imread, data.py:38) when a dataset would sample an HR frame that was not on disk."""
        have = getattr(self, 'have_hr', None)
        if have is not None:
            missing = [t for t in frames if not have[t]]
            if missing:
                raise FileNotFoundError('HR frame(s) missing: ' + ', '.join(f'frame_{t:05d}.png' for t in missing[:8]))


def _store_for(opt):
    store = getattr(opt, 'frame_store', None)
    if store is None:
        store = FrameStore.from_directory(opt.dataset, opt.scene)
        opt.frame_store = store            # decoded once, shared by the three datasets
    return store


class VideoDataset(torch.utils.data.Dataset):
    """Base class (reference data.py:14-45): a list of centre-frame indices over a FrameStore."""

    def __init__(self, opt, transform=None):
        self.fps, self.win_size, self.transform = opt.fps, opt.lr_window, transform
        self.store = _store_for(opt)
        self.frames = []
        self.populate_files(self.store.num_lr, opt)
        self.store.require_hr(self.frames)

    def __len__(self):
        return len(self.frames)

    def batch(self, positions):
        """dict(hr=(n,3,H,W), lr=(n,(2w+1)*4,h,w)) for dataset positions, produced on the store's device."""
        idx = torch.tensor([self.frames[int(p)] for p in positions], dtype=torch.int32)
        if self.store.device.type == 'cuda':
            # pinned + non_blocking: a pageable H2D copy is synchronous, i.e. the host would wait for the previous step's last
            # kernel on this stream before it may enqueue the next step (the host then never runs ahead of the GPU)
            idx = idx.pin_memory().to(self.store.device, non_blocking=True)
        hr, lr = sample_windows(self.store.hr, self.store.lr, idx, self.win_size)
        sample = {'hr': hr, 'lr': lr}
        return self.transform(sample) if self.transform else sample

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        return {k: v[0] for k, v in self.batch([idx]).items()}


class VideoTrainDataset(VideoDataset):
    """Sparse HR frames with their LR windows."""

    def __init__(self, opt, transform=None):
        super().__init__(opt, transform)
        self.shuffle = True

    def populate_files(self, num_lr, opt):
        self.frames = list(range(1 + opt.fps, num_lr - opt.fps, 120 // opt.fps))


class VideoAllDataset(VideoDataset):
    """Every LR window (used for HR generation after training)."""

    def __init__(self, opt, transform=None):
        super().__init__(opt, transform)
        self.shuffle = opt.operation == 'train'

    def populate_files(self, num_lr, opt):
        self.frames = list(range(1 + opt.fps, num_lr - opt.fps))


class VideoValDataset(VideoDataset):
    """k uniformly drawn frames that are not training frames."""

    def __init__(self, opt, k, transform=None):
        self.k = k
        super().__init__(opt, transform)
        self.shuffle = False

    def populate_files(self, num_lr, opt):
        for i in torch.randperm(num_lr - 2 * opt.lr_window).tolist():
            i += opt.lr_window
            if (i + opt.fps + 3) % (120 // opt.fps) == 0:
                continue
            self.frames.append(i)
            if len(self.frames) == self.k:
                break


class ConcatDataset(torch.utils.data.Dataset):
    """Supervised sample i paired with a uniformly random unsupervised sample."""

    def __init__(self, sup, unsup):
        self.sup, self.unsup = sup, unsup
        self.num_sup, self.num_unsup = len(sup), len(unsup)

    def __len__(self):
        return self.num_sup

    def pair_positions(self, positions):
        return [torch.randint(self.num_unsup, (1, 1)).item() for _ in positions]

    def batch(self, positions):
        return [self.sup.batch(positions), self.unsup.batch(self.pair_positions(positions))]

    def __getitem__(self, i):
        return self.sup[i], self.unsup[self.pair_positions([i])[0]]


class DeviceLoader:
    """Iterates a dataset in batches built on the GPU by the sampler kernel (stands in for torch DataLoader +
    default_collate + H2D copy).  Under data parallel each rank walks positions rank::world of the order padded to a
    multiple of the world size."""

    def __init__(self, dataset, batch_size, shuffle=False, device=None):
        self.dataset, self.batch_size, self.shuffle, self.device = dataset, batch_size, shuffle, device

    def _store(self):
        ds = self.dataset.sup if isinstance(self.dataset, ConcatDataset) else self.dataset
        return ds.store

    def _per_rank(self):
        _, ws = sdist.world()
        return (len(self.dataset) + ws - 1) // ws

    def __len__(self):
        return (self._per_rank() + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        store = self._store()
        if store.device.type != 'cuda':
            dev = self.device or torch.device('cuda', torch.cuda.current_device())
            store.to(dev)
        rank, ws = sdist.world()
        order = torch.randperm(len(self.dataset)).tolist() if self.shuffle else list(range(len(self.dataset)))
        # DistributedSampler semantics (what Lightning DDP gives the reference): pad by wrapping around so that every rank
        # walks the same number of positions -- same number of batches and the same last-batch size on every rank, hence
        # the same number of gradient all-reduces (a rank with one batch more would hang in the collective)
        total = self._per_rank() * ws
        if order:
            order = (order * (total // len(order) + 1))[:total]
        order = order[rank::ws]
        for s in range(0, len(order), self.batch_size):
            yield self.dataset.batch(order[s:s + self.batch_size])


def get_loader(dataset, batch=4):
    return DeviceLoader(dataset, batch, shuffle=dataset.shuffle)


class LitTrainLoader(pl.LightningDataModule):
    def __init__(self, train_data, val_data, batch):
        super().__init__()
        self.batch, self.train_data, self.val_data = batch, train_data, val_data

    def train_dataloader(self):
        return DeviceLoader(self.train_data, self.batch)          # the reference never shuffles here (data.py:134)

    def val_dataloader(self):
        return DeviceLoader(self.val_data, 40)                    # fixed 40 (data.py:137)
