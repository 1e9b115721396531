"""Data path either side of the training step: src/dtk.jl, src/kitty.jl, src/dchain.jl, src/io_utils.jl and the
`train()` loop of src/Monodepth.jl.  The host logic runs on the CPU; the loop itself needs the HIP library (gpu marker)."""
import os

import numpy as np
import pytest

import md2amd

D = md2amd.data
Image = pytest.importorskip("PIL.Image")


def _strip(path, seed, gray=False, static=False):
    x = md2amd.synthetic.synthetic_triplets(1, 1 if gray else 3, 128, 416, seed=seed)[0]          # (3, C, H, W)
    if static:
        x[0] = x[1]; x[2] = x[1]
    frames = [np.transpose(f, (1, 2, 0)) for f in x]
    strip = np.concatenate(frames, 1)
    u8 = (strip * 255.0 + 0.5).astype(np.uint8)
    Image.fromarray(u8[..., 0] if gray else u8).save(path)
    return u8


def test_depth10k_splits_the_strip_and_builds_the_intrinsics(tmp_path):
    u8 = _strip(tmp_path / "a.png", 1)
    ds = D.Depth10k(str(tmp_path), ["a.png"])
    assert len(ds) == 1 and ds.resolution == (416, 128) and ds.source_ids == [1, 3] and ds.target_id == 2
    f = 2648.0 / 4.63461538462
    assert np.allclose(ds.K, [[f, 0, 208], [0, f, 64], [0, 0, 1]]) and np.allclose(ds.K @ ds.invK, np.eye(3), atol=1e-12)
    x = ds[0]
    assert x.shape == (3, 3, 128, 416) and x.dtype == np.float32
    for j in range(3):                                      # frame j = columns [416 j, 416 (j+1)), channels first, N0f8 / 255
        assert np.array_equal(x[j], np.transpose(u8[:, 416 * j:416 * (j + 1)].astype(np.float32) / np.float32(255), (2, 0, 1)))
    g = D.Depth10k(str(tmp_path), ["a.png"], grayscale=True)[0]
    assert g.shape == (3, 1, 128, 416)
    assert np.allclose(g[:, 0], 0.299 * x[:, 0] + 0.587 * x[:, 1] + 0.114 * x[:, 2], atol=1e-6)


def test_flipx_mirrors_every_frame_together(tmp_path):
    _strip(tmp_path / "a.png", 2)
    plain = D.Depth10k(str(tmp_path), ["a.png"])[0]
    assert np.array_equal(D.Depth10k(str(tmp_path), ["a.png"], augmentations=D.FlipX(1.0))[0], plain[..., ::-1])
    assert np.array_equal(D.Depth10k(str(tmp_path), ["a.png"], augmentations=D.FlipX(0.0))[0], plain)
    flips = [np.array_equal(D.FlipX(0.5, seed=s)([plain[0, 0]])[0], plain[0, 0, :, ::-1]) for s in range(40)]
    assert 5 < sum(flips) < 35


def _kitti(root, seq, n_frames, size=(94, 310)):
    d = root / "sequences" / seq / "image_0"
    d.mkdir(parents=True)
    (root / "sequences" / seq / "calib.txt").write_text(
        "P0: 718.856 0.0 607.1928 0.0 0.0 718.856 185.2157 0.0 0.0 0.0 1.0 0.0\nP1: 1 0 0 0 0 1 0 0 0 0 1 0\n")
    rng = np.random.default_rng(7)
    imgs = []
    for i in range(n_frames):
        a = (rng.random(size) * 255).astype(np.uint8)
        Image.fromarray(a).save(d / ("%06d.png" % i)); imgs.append(a)
    return imgs


def test_kitty_dataset_triplets_intrinsics_and_resize(tmp_path):
    imgs = _kitti(tmp_path, "00", 8)
    ds = D.KittyDataset(str(tmp_path), "00", target_size=(32, 104))
    assert len(ds) == 2 and ds.resolution == (104, 32) and ds.frame_ids == [1, 2, 3]      # 8 frames -> 2 non-overlapping triplets
    fx = np.mean([32 / 94, 104 / 310]) * 718.856
    assert np.allclose(ds.K, [[fx, 0, 52], [0, fx, 16], [0, 0, 1]])
    x = ds[1]                                               # frames 3, 4, 5
    assert x.shape == (3, 1, 32, 104)
    for j in range(3):
        assert np.allclose(x[j, 0], D.imresize(imgs[3 + j].astype(np.float32) / 255, (32, 104)), atol=1e-6)
    assert np.allclose(D.parse_matrix("1 2 3 4 5 6 7 8 9 10 11 12"), [[1, 2, 3], [5, 6, 7], [9, 10, 11]])


def test_imresize_is_linear_and_centre_aligned():
    a = np.random.default_rng(0).random((6, 10)).astype(np.float32)
    assert np.allclose(D.imresize(a, (6, 10)), a)                                           # same size: identity
    ramp = np.tile(np.arange(12, dtype=np.float32), (4, 1))                                 # linear in x: stays linear away from the border clamp
    out = D.imresize(ramp, (4, 6))
    assert np.allclose(out[0], (np.arange(6) + 0.5) * 2 - 0.5)
    rgb = np.stack([a, 2 * a, 3 * a], -1)
    assert np.allclose(D.imresize(rgb, (3, 5))[..., 2], 3 * D.imresize(a, (3, 5)), atol=1e-6)


class _Const:
    def __init__(self, n, v): self.n, self.v = n, v
    def __len__(self): return self.n
    def __getitem__(self, i): return np.full((3, 1, 2, 2), self.v * 100 + i, np.float32)


def test_dchain_maps_indices_to_datasets():
    c = D.DChain([_Const(2, 1), _Const(3, 2), _Const(1, 3)])
    assert len(c) == 6 and c.bins == [2, 5, 6]
    assert [int(c[i][0, 0, 0, 0]) for i in range(6)] == [100, 101, 200, 201, 202, 300]
    with pytest.raises(IndexError):
        c[6]


def test_dataloader_full_batches_shuffle_and_buffers():
    c = D.DChain([_Const(7, 1)])
    plain = [b.numpy().copy() for b in D.DataLoader(c, 3, pin_memory=False)]
    assert len(plain) == 2 and plain[0].shape == (3, 3, 1, 2, 2)                            # 7 // 3 full batches, the remainder is dropped
    assert [int(v) for b in plain for v in b[:, 0, 0, 0, 0]] == [100, 101, 102, 103, 104, 105]
    s1 = [int(v) for b in D.DataLoader(c, 3, shuffle=True, seed=5, pin_memory=False) for v in b[:, 0, 0, 0, 0]]
    s2 = [int(v) for b in D.DataLoader(c, 3, shuffle=True, seed=5, pin_memory=False) for v in b[:, 0, 0, 0, 0]]
    assert s1 == s2 and len(set(s1)) == 6 and set(s1) <= set(range(100, 107)) and s1 != [100, 101, 102, 103, 104, 105]


def test_png_writers(tmp_path):
    d = np.linspace(0, 1, 32 * 48, dtype=np.float32).reshape(32, 48)
    D.save_disparity(d, tmp_path / "d.png"); D.save_warped(d[None], tmp_path / "w1.png"); D.save_warped(np.stack([d, d, d]), tmp_path / "w3.png")
    a = np.asarray(Image.open(tmp_path / "d.png")); w1 = np.asarray(Image.open(tmp_path / "w1.png")); w3 = np.asarray(Image.open(tmp_path / "w3.png"))
    assert a.shape == (32, 48, 3) and w1.shape == (32, 48) and w3.shape == (32, 48, 3)
    assert tuple(a[0, 0]) != tuple(a[-1, -1]) and w1[0, 0] == 0 and w1[-1, -1] == 255
    assert np.allclose(D.thermal(np.array([0.0, 1.0])), [[0.016, 0.138, 0.202], [0.910, 0.981, 0.355]], atol=1e-6)


@pytest.mark.gpu
def test_train_loop_and_find_static_on_a_tiny_depth10k(tmp_path):
    import torch
    files = []
    for i in range(5):
        _strip(tmp_path / ("s%d.png" % i), 10 + i, gray=True, static=(i == 4)); files.append("s%d.png" % i)
    ds = D.Depth10k(str(tmp_path), files, augmentations=D.FlipX(0.5, seed=1), grayscale=True)
    T = md2amd.train
    model, losses = T.train([ds], str(tmp_path / "log"), str(tmp_path / "ckpt"), batch_size=2, n_epochs=2, log_iter=2, save_iter=2, quiet=True)
    assert len(losses) == 4 and all(np.isfinite(losses)) and all(0 < l < 1 for l in losses)
    assert os.path.exists(tmp_path / "log" / "disp-1-1.png") and os.path.exists(tmp_path / "log" / "warp-1-2-2.png")
    assert any(n.startswith("1-2-") and n.endswith(".npz") for n in os.listdir(tmp_path / "ckpt"))
    w = np.asarray(Image.open(tmp_path / "log" / "warp-1-1-1.png"))
    assert w.shape == (128, 416) and w.std() > 1                                            # a warped source image, not a blank
    # find_static (src/dtk.jl:51-69): the sample whose three frames are identical has zero automasking loss
    plain = D.Depth10k(str(tmp_path), files, grayscale=True)
    assert D.find_static(plain, 1e-4) == files[:4]
    torch.cuda.synchronize()
