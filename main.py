"""Command line entry of the sin-inn path, MI355X build (drop-in for the reference's main.py): same sub-commands
(``train`` / ``test``) and the same flags (reference main.py:9-83), driving the HIP-backed ``SingleVideoINN``.

Differences from the reference, all to make the path runnable here: the trainer / logger come from
``sin_inn_amd.lightning`` (pytorch_lightning 1.2 and wandb are not installable), devices are taken from ``--gpu_ids``
without hard-coded 'cuda' strings, ``--tcr_iters`` is an int, ``--synthetic T H W`` trains on a synthetic clip
(no dataset directory needed), and under ``torchrun`` every rank trains on its shard with one RCCL gradient
all-reduce per step.
"""
import argparse
import logging
import os
import re

import torch

from data import (ConcatDataset, FrameStore, LitTrainLoader, VideoAllDataset, VideoTrainDataset, VideoValDataset,
                  get_loader)
from lit_wrapper import SingleVideoINN
from sin_inn_amd.lightning import ModelCheckpoint, Trainer, WandbLogger, load_checkpoint

# (flags, kwargs) -- one row per option of the reference CLI
_OPTIONS = [
    (('-g', '--gpu_ids'), dict(nargs='+', type=int, default=[0], help='GPU ids to use')),
    (('--dataset',), dict(default='datasets/adobe240f', help='dataset root (hr_frames/ and lr_frames/ inside)')),
    (('-s', '--scene'), dict(default='IMG_0028_binning_4x', help='video (sub-directory) name')),
    (('--suffix',), dict(default='default', help='experiment name suffix')),
    (('-f', '--fps'), dict(type=int, default=10, help='HR frame rate; LR frames are 120 fps')),
    (('--lr_window',), dict(type=int, default=10, help='LR frames taken on each side of an HR frame')),
    (('-b', '--batch_size'), dict(type=int, default=8, help='batch size per GPU')),
    (('-a', '--architecture'), dict(choices=['SRF', 'IRN'], default='SRF')),
    (('--scale',), dict(type=int, default=4, help='HR / LR resolution ratio')),
    (('-c', '--num_coupling'), dict(type=int, default=4, help='GLOW blocks between downsamples')),
    (('-r', '--resume_state'), dict(default=None, help='checkpoint to resume from / to test')),
    (('-w', '--working_dir'), dict(default='experiments', help='where logs and checkpoints go')),
    (('-e', '--epochs'), dict(type=int, default=10000)),
    (('--save_iter',), dict(type=int, default=100, help='checkpoint every N epochs')),
    (('-p', '--print_iter'), dict(type=int, default=10, help='validate / log every N epochs')),
    (('-l', '--learning_rate'), dict(type=float, default=1e-4)),
    (('--adam_betas',), dict(type=float, nargs=2, default=[0.9, 0.99])),
    (('--weight_decay',), dict(type=float, default=1e-5)),
    (('--lambda_fwd_rec',), dict(type=float, default=1)),
    (('--lambda_fwd_mmd',), dict(type=float, default=0)),
    (('--lambda_latent_nll',), dict(type=float, default=0)),
    (('--lambda_bwd_rec',), dict(type=float, default=1)),
    (('--lambda_bwd_mmd',), dict(type=float, default=0)),
    (('--random_seed',), dict(type=int, default=0)),
    (('--lambda_bwd_tcr',), dict(type=float, default=0)),
    (('--rotation',), dict(type=float, default=5, help='TCR rotation range, degrees')),
    (('--translation',), dict(type=float, default=5, help='TCR translation range, pixels')),
    (('--tcr_iters',), dict(type=int, default=5, help='TCR samples per image')),
    (('-t', '--temp',), dict(type=float, default=0.8, help='latent sampling temperature at test time')),
    (('--lr_dims',), dict(type=int, default=-1, help='internal: LR channels')),
    (('--z_dims',), dict(type=int, default=-1, help='internal: latent channels')),
    (('--precision',), dict(choices=['fp32', 'bf16'], default='fp32',
                            help='fp32: the reference arithmetic; bf16: conv subnets on bf16 MFMA (fp32 accumulate), fp32 flow')),
    (('--pixel_mode',), dict(choices=['clamp', 'wrap'], default='clamp',
                             help='test: float->uint8 conversion; wrap = the reference ToPILImage wrap-around')),
    (('--save_images',), dict(default=None, help='test: write PNG frames to this directory instead of the ffmpeg pipe')),
    (('--allow_partial_load',), dict(action='store_true', help='test: load a checkpoint whose keys do not all match')),
    (('--trust_checkpoint',), dict(action='store_true', help='load --resume_state with the full unpickler (it can run code from '
                                                             'the file); default: tensors, primitives and argparse.Namespace only')),
    (('--synthetic',), dict(type=int, nargs=3, default=None, metavar=('T', 'H', 'W'),
                            help='use a synthetic uint8 clip of T frames of HxW instead of --dataset')),
]


def get_args(argv=None):
    ap = argparse.ArgumentParser(description='Train an invertible network on a single video (MI355X build)')
    ap.add_argument('operation', choices=['train', 'test'])
    for flags, kw in _OPTIONS:
        ap.add_argument(*flags, **kw)
    args = ap.parse_args(argv)
    args.lr_dims = (2 * args.lr_window + 1) * 4
    args.z_dims = args.scale * args.scale * 3 * 4 - args.lr_dims
    logging.basicConfig(level=logging.INFO)
    torch.manual_seed(args.random_seed)
    assert args.scale % 4 == 0
    assert args.z_dims >= 0, 'lr_window too large for this scale'
    if args.operation == 'test':
        assert args.resume_state is not None and os.path.isfile(args.resume_state), \
            'Please provide weights using --resume_state'
    if args.synthetic is not None:
        t, h, w = args.synthetic
        args.frame_store = FrameStore.synthetic(t, h, w)
    return args


# state-dict entries that differ between FrEIA versions without changing the function: the fixed permutations (derived
# from the seed, archs.py:65-68) and bookkeeping tensors some versions register (github.com/VLL-HD/FrEIA/issues/10, the
# mismatch the reference asks about interactively at main.py:128-136)
_FREIA_BOOKKEEPING = re.compile(r'\.(perm|perm_inv|w_perm|w_perm_inv|last_jac|tmp_var\d*)$')


def load_weights(model, state_dict, allow_partial=False):
    """Strict load; a mismatch is tolerated only when every offending key is FrEIA bookkeeping (or --allow_partial_load
    is given).  The reference stops and asks (main.py:128-136); a batch job cannot, so anything else exits non-zero
    instead of silently running inference with random weights."""
    try:
        model.load_state_dict(state_dict)
        return
    except RuntimeError as e:
        logging.warning(str(e))
    res = model.load_state_dict(state_dict, strict=False)
    odd = [k for k in list(res.missing_keys) + list(res.unexpected_keys) if not _FREIA_BOOKKEEPING.search(k)]
    if odd and not allow_partial:
        logging.error(f'checkpoint does not match the model ({len(odd)} keys, e.g. {odd[:4]}); '
                      'pass --allow_partial_load to load what matches anyway')
        raise SystemExit(1)


def main(argv=None):
    args = get_args(argv)
    sup_data = VideoTrainDataset(args)
    unsup_data = VideoAllDataset(args)
    train_data = ConcatDataset(sup_data, unsup_data)
    val_data = VideoValDataset(args, len(train_data) * 4 // 6)

    # image dimensions come from the frame store (the reference decodes one batch for this, main.py:96-98)
    _, height, width, channels = unsup_data.store.hr.shape
    model = SingleVideoINN(channels, height, width, args)

    if args.operation == 'train':
        exp_dir = os.path.join(args.working_dir, args.operation, f'{args.scene}_{args.architecture}_{args.suffix}')
        os.makedirs(exp_dir, exist_ok=True)
        logger = WandbLogger(project='sin-inn', save_dir=exp_dir, name=os.path.basename(exp_dir))
        logger.log_hyperparams(argparse.Namespace(**{k: v for k, v in vars(args).items() if k != 'frame_store'}))
        trainer = Trainer(check_val_every_n_epoch=args.print_iter, default_root_dir=exp_dir, gpus=args.gpu_ids,
                          logger=logger, max_epochs=args.epochs, resume_from_checkpoint=args.resume_state,
                          callbacks=[ModelCheckpoint(period=args.save_iter)], trust_checkpoint=args.trust_checkpoint)
        trainer.fit(model, LitTrainLoader(train_data, val_data, args.batch_size))
    else:
        exp_dir = os.path.join(args.working_dir, args.operation, args.scene)
        os.makedirs(exp_dir, exist_ok=True)
        video_path = os.path.join(exp_dir, f'{args.architecture}_{args.suffix}_t{args.temp}.avi')
        device = torch.device('cuda', args.gpu_ids[0])
        checkpoint = load_checkpoint(args.resume_state, map_location=device, trust=args.trust_checkpoint)
        load_weights(model, checkpoint['state_dict'], args.allow_partial_load)
        model.to(device)
        if args.save_images:
            model.infer(get_loader(unsup_data, 40), args, save_images=args.save_images)
        else:
            model.infer(get_loader(unsup_data, 40), args, save_video=video_path)
    return model


if __name__ == '__main__':
    main()
