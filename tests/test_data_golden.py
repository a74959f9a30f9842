"""CPU: the oracle AND the host-side product logic (data.py / main.py / tcr.py of this repository) against fixtures G8-G11,
which tests/golden/make_golden_data.py produced by running the reference's own prepare.py / data.py / main.py / tcr.py."""
import math
import types

import numpy as np
import pytest
import torch

from oracle import sininn_oracle as O

CASES = range(6)


# ---- G8: datasets/prepare.py:35-82,103-116,164 ---------------------------------------------------------------------------
@pytest.mark.parametrize('reduction', ['mean', 'sum'])
@pytest.mark.parametrize('scale', [1, 2, 4])
def test_g8_bayer_bin_oracle_is_byte_exact_with_the_reference(golden_data, scale, reduction):
    frames = golden_data['g8_frames']
    want = np.stack([golden_data[f'g8_lr_u8_{t}_{reduction}_{scale}'] for t in range(len(frames))])
    got = O.bayer_bin(frames, scale, reduction)
    assert got.dtype == np.uint8 and np.array_equal(got, want)
    for t, frame in enumerate(frames):
        mosaic, planes = O.bayer_planes(frame, scale, reduction)
        assert np.array_equal(mosaic, golden_data[f'g8_bayer_{t}'])                                # RGGB sampling, :35-52
        assert np.array_equal(np.stack(planes, -1), golden_data[f'g8_binned_{t}_{reduction}_{scale}'])   # float64, bit-exact
        assert np.array_equal(O.bayer_mosaic(frame, scale, reduction), golden_data[f'g8_cfa_{t}_{reduction}_{scale}'])


def test_g8_fixture_exercises_clipping_and_floor(golden_data):
    assert golden_data['g8_binned_1_sum_4'].max() > 1.0 and golden_data['g8_lr_u8_1_sum_4'].max() == 255
    small = golden_data['g8_binned_2_mean_4'][:, :2] * 255
    assert (small - np.floor(small)).max() > 0.5          # values whose rounding would differ from truncation


# ---- G9: data.py:22,55-59,72-76,87-99,105-118 ----------------------------------------------------------------------------
def _opt(listed, fps, win, operation='train'):
    from data import FrameStore
    store = FrameStore(torch.zeros(listed, 8, 8, 3, dtype=torch.uint8), torch.zeros(listed, 1, 1, 4, dtype=torch.uint8),
                       num_listed=listed)
    return types.SimpleNamespace(fps=fps, lr_window=win, operation=operation, frame_store=store)


@pytest.mark.parametrize('ci', CASES)
def test_g9_index_arithmetic_oracle_and_datasets(golden_data, ci):
    from data import ConcatDataset, VideoAllDataset, VideoTrainDataset, VideoValDataset
    listed, fps, win, k = (int(v) for v in golden_data['g9_cases'][ci])
    num_lr = listed - 1
    train, every, val = (golden_data[f'g9_{ci}_{n}'].tolist() for n in ('train', 'all', 'val'))
    # oracle restatement
    assert O.train_indices(num_lr, fps) == train and O.all_indices(num_lr, fps) == every
    torch.manual_seed(100 + ci)
    assert O.val_indices(num_lr, fps, win, k, torch.randperm(num_lr - 2 * win).tolist()) == val
    # host-side product logic
    opt = _opt(listed, fps, win)
    sup, unsup = VideoTrainDataset(opt), VideoAllDataset(opt)
    assert sup.frames == train and unsup.frames == every and len(sup) == len(train)
    torch.manual_seed(100 + ci)
    vds = VideoValDataset(opt, k)
    assert vds.frames == val
    assert [sup.shuffle, unsup.shuffle, vds.shuffle] == golden_data[f'g9_{ci}_shuffle'].tolist()
    assert VideoAllDataset(_opt(listed, fps, win, 'test')).shuffle == bool(golden_data[f'g9_{ci}_all_shuffle_test'])
    # the LR window of a sample: frames idx-win .. idx+win (data.py:57-58)
    if train:
        assert golden_data[f'g9_{ci}_train_window0'].tolist() == list(range(train[0] - win, train[0] + win + 1))
    # supervised / unsupervised pairing: same length, same random stream (data.py:112-115)
    cd = ConcatDataset(sup, unsup)
    pairs = golden_data[f'g9_{ci}_pairs']
    assert len(cd) == int(golden_data[f'g9_{ci}_len']) == len(pairs)
    torch.manual_seed(200 + ci)
    assert cd.pair_positions(range(len(cd))) == pairs[:, 1].tolist()


def test_g9_quirks_are_in_the_fixture(golden_data):
    assert len(golden_data['g9_4_val']) == 59           # k = 0: `num == k` never fires, the whole permutation is taken (C-10)
    assert len(golden_data['g9_5_val']) == 0            # fps 120: every frame is a training frame
    assert golden_data['g9_2_train'].tolist() == [2]    # BASELINE configs[0]: 8 entries, fps 1, lr_window 1


# ---- G10: main.py:9-83 ---------------------------------------------------------------------------------------------------
def test_g10_get_args_dims_and_defaults(golden_data):
    import main
    for ai in range(5):
        argv = golden_data[f'g10_{ai}_argv'].tolist()
        if '--scale' in argv:
            continue                                    # scale 8 passes the reference's assert but no network has those dims (C-11)
        a = main.get_args(argv)
        assert [a.lr_dims, a.z_dims] == golden_data[f'g10_{ai}_dims'].tolist()
    a = vars(main.get_args(['train']))
    names, values, types_ = (golden_data[f'g10_default_{n}'].tolist() for n in ('names', 'values', 'types'))
    for name, value, tp in zip(names, values, types_):
        assert name in a, name
        if name == 'tcr_iters':                         # the reference's float makes range() raise (C-4); an int here
            assert a[name] == 5 and value == '5'
            continue
        assert repr(a[name]) == value and type(a[name]).__name__ == tp, name


# ---- G11: tcr.py:26-45 (everything the reference computes itself; kornia's warp stays unpinned) ----------------------------
@pytest.mark.parametrize('ti', range(3))
def test_g11_tcr_rotation_inputs_and_translation(golden_data, ti):
    import tcr
    b, c, h, w, ang, trans, scale = golden_data[f'g11_{ti}_cfg'].tolist()
    b, h, w = int(b), int(h), int(w)
    rand = torch.from_numpy(golden_data[f'g11_{ti}_rand'])
    angle = torch.from_numpy(golden_data[f'g11_{ti}_angle'])
    shift = torch.from_numpy(golden_data[f'g11_{ti}_translation'])
    assert golden_data[f'g11_{ti}_center'].tolist() == [[w / 2, h / 2]] * b and golden_data[f'g11_{ti}_dsize'].tolist() == [h, w]
    assert np.all(golden_data[f'g11_{ti}_zoom'] == 1)
    for mat in (O.tcr_matrix(rand, h, w, ang, trans, scale), tcr.pixel_matrix(rand, h, w, ang, trans, scale)):
        rad = angle * (math.pi / 180)
        cos, sin = torch.cos(rad), torch.sin(rad)
        assert torch.allclose(mat[:, 0, 0], cos, atol=1e-6) and torch.allclose(mat[:, 0, 1], sin, atol=1e-6)
        rot_shift = torch.stack([(1 - cos) * (w / 2) - sin * (h / 2), sin * (w / 2) + (1 - cos) * (h / 2)], 1)
        assert torch.allclose(mat[:, :, 2] - rot_shift, shift, atol=2e-5)
    if scale != 1:
        assert shift.abs().max() > trans                # C-6: dividing by scale = 1/4 makes the LR shift 4x the HR one
