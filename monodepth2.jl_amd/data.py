"""Data path either side of the training step: datasets, chaining, flip augmentation, a loader that fills
pinned host batches, `find_static`, and the PNG writers of the training loop.  Host-side Python (the reference's is host-side Julia);
nothing here is on the measured path — bench.py keeps its synthetic, HBM-resident input.

Every sample is a float32 array (3, C, H, W) in [0,1] — frames (source 1, target 2, source 3) — which is the reference's
(W, H, C, 3) column-major sample seen row-major; a batch is (N, 3, C, H, W), the layout `Model`/`train_loss` take.

Reference: src/dtk.jl:1-69 (Depth10k, find_static), src/kitty.jl:1-100 (KittyDataset), src/dchain.jl:1-30 (DChain),
src/Monodepth.jl:84,156-160 (FlipX(0.5), DataLoader(shuffleobs(dchain), batch_size)), src/io_utils.jl:1-24 (save_disparity / save_warped).

Third-party semantics restated here and NOT verifiable in this environment (no Julia): Images.jl `load` + `Gray{Float32}` conversion
(ITU-R BT.601 luma 0.299 R + 0.587 G + 0.114 B on N0f8 values / 255), ImageTransformations `imresize` (linear interpolation on
pixel-centre-aligned coordinates, no antialiasing prefilter), Augmentations.jl `FlipX(p)` (one coin per sample, all frames flipped
together), MLDataPattern `shuffleobs`, and the `:thermal` colour map of Plots.jl (approximated by a 5-stop ramp).
"""
import os

import numpy as np

try:                                    # Pillow decodes / encodes PNG; present in this image and on the GPU box
    from PIL import Image
except Exception:                       # pragma: no cover
    Image = None


# ---------------------------------------------------------------------------------------------- image helpers
def load_image(path):
    """Images.load: (H, W) uint8 for grayscale files, (H, W, 3) uint8 for colour files."""
    if Image is None:
        raise RuntimeError("Pillow is required to read image files")
    with Image.open(path) as im:
        if im.mode in ("L", "I;16", "I", "1"):
            a = np.asarray(im.convert("L"))
        else:
            a = np.asarray(im.convert("RGB"))
    return a


def to_float(a):
    """N0f8 -> Float32."""
    return a.astype(np.float32) / np.float32(255.0)


def to_gray(rgb):
    """Gray{Float32}.(img) (src/dtk.jl:33): BT.601 luma of float RGB (H, W, 3) -> (H, W)."""
    if rgb.ndim == 2:
        return rgb
    return (np.float32(0.299) * rgb[..., 0] + np.float32(0.587) * rgb[..., 1] + np.float32(0.114) * rgb[..., 2]).astype(np.float32)


def imresize(img, size):
    """imresize(img, (height, width)) (src/kitty.jl:60): linear interpolation, output pixel centre o maps to input coordinate
    (o + 0.5) * in/out - 0.5, clamped to the image; channels (last axis, if any) are independent."""
    h, w = size
    H, W = img.shape[:2]
    a = img.astype(np.float32)

    def axis(n_out, n_in):
        c = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        c = np.clip(c, 0.0, n_in - 1.0)
        i0 = np.floor(c).astype(np.int64)
        i1 = np.minimum(i0 + 1, n_in - 1)
        return i0, i1, (c - i0).astype(np.float32)

    r0, r1, fr = axis(h, H)
    c0, c1, fc = axis(w, W)
    fr = fr.reshape((-1, 1) + (1,) * (a.ndim - 2)); fcb = fc.reshape((1, -1) + (1,) * (a.ndim - 2))
    top = a[r0][:, c0] * (1 - fcb) + a[r0][:, c1] * fcb
    bot = a[r1][:, c0] * (1 - fcb) + a[r1][:, c1] * fcb
    return (top * (1 - fr) + bot * fr).astype(np.float32)


class FlipX:
    """Augmentations.FlipX(p) (src/Monodepth.jl:84): with probability p mirror every frame of the sample along the width."""

    def __init__(self, p=0.5, seed=0):
        self.p = p
        self.rng = np.random.default_rng(seed)

    def __call__(self, frames):
        if self.rng.random() < self.p:
            return [np.ascontiguousarray(f[:, ::-1]) for f in frames]
        return frames


def _stack(frames):
    """list of 3 (H, W) or (H, W, 3) float frames -> (3, C, H, W)."""
    out = []
    for f in frames:
        out.append(f[None] if f.ndim == 2 else np.transpose(f, (2, 0, 1)))
    return np.ascontiguousarray(np.stack(out, 0), dtype=np.float32)


def _intrinsics(fx, fy, cx, cy):
    K = np.array([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]], np.float64)
    return K, np.linalg.inv(K)


# ---------------------------------------------------------------------------------------------- datasets
class Depth10k:
    """src/dtk.jl:1-46 — every file is a horizontal strip of 3 frames of 416x128."""

    def __init__(self, image_dir, image_files, augmentations=None, grayscale=False):
        focal = 2648.0 / 4.63461538462
        self.resolution = (416, 128)                        # (width, height)
        self.K, self.invK = _intrinsics(focal, focal, self.resolution[0] / 2.0, self.resolution[1] / 2.0)
        self.dir, self.files = image_dir, list(image_files)
        self.source_ids, self.target_id = [1, 3], 2
        self.augmentations, self.grayscale = augmentations, grayscale

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        width = self.resolution[0]
        strip = to_float(load_image(os.path.join(self.dir, self.files[i])))
        if self.grayscale:
            strip = to_gray(strip)
        elif strip.ndim == 2:
            strip = np.repeat(strip[..., None], 3, -1)
        if strip.shape[1] < 3 * width:
            raise ValueError("%s: expected a strip of 3 frames of width %d" % (self.files[i], width))
        frames = [strip[:, width * j:width * (j + 1)] for j in range(3)]
        if self.augmentations is not None:
            frames = self.augmentations(frames)
        return _stack(frames)


def parse_matrix(line):
    """src/kitty.jl:77-81: 12 numbers of a calib.txt row -> the 3x3 block of the (row-major) 3x4 projection matrix."""
    m = np.array([float(t) for t in line.split(" ") if t], np.float64)
    if m.size != 12:
        raise ValueError("calib line must hold 12 numbers")
    return m.reshape(3, 4)[:, :3].copy()


class KittyDataset:
    """src/kitty.jl:1-75 — odometry sequence `image_dir/sequences/<seq>/image_0/%06d.png`, non-overlapping triplets of consecutive
    frames resized to `target_size` = (height, width); fx = mean(target/original) * K[1,1], principal point at the image centre."""

    def __init__(self, image_dir, sequence, target_size, augmentations=None):
        seq_dir = os.path.join(image_dir, "sequences", sequence)
        with open(os.path.join(seq_dir, "calib.txt")) as f:
            first = f.readline().rstrip("\n")
        K0 = parse_matrix(first[4:])                        # drop the "P0: " tag (Ks[1][5:end])
        self.frames_dir = os.path.join(seq_dir, "image_0")
        files = sorted(n for n in os.listdir(self.frames_dir) if n.endswith(".png"))
        if not files:
            raise ValueError("no frames under " + self.frames_dir)
        original = load_image(os.path.join(self.frames_dir, files[0])).shape[:2]
        height, width = target_size
        fx = float(np.mean([height / original[0], width / original[1]])) * K0[0, 0]
        self.K, self.invK = _intrinsics(fx, fx, width // 2, height // 2)
        self.resolution = (width, height)
        self.source_ids, self.target_id, self.frame_ids = [1, 3], 2, [1, 2, 3]
        self.total_length = len(files) // len(self.frame_ids)
        self.augmentations = augmentations

    def __len__(self):
        return self.total_length

    def __getitem__(self, i):
        sid = i * len(self.frame_ids)
        width, height = self.resolution
        frames = []
        for x in self.frame_ids:
            im = to_gray(to_float(load_image(os.path.join(self.frames_dir, "%06d.png" % (sid + x - 1)))))
            frames.append(imresize(im, (height, width)))
        if self.augmentations is not None:
            frames = self.augmentations(frames)
        return _stack(frames)


class DChain:
    """src/dchain.jl — concatenation of datasets: index i falls into the first dataset whose cumulative length exceeds it."""

    def __init__(self, datasets):
        self.datasets = list(datasets)
        self.bins = np.cumsum([len(d) for d in self.datasets]).tolist()

    def __len__(self):
        return self.bins[-1] if self.bins else 0

    def __getitem__(self, i):
        if i < 0 or i >= len(self):
            raise IndexError(i)
        bid = 0
        while i >= self.bins[bid]:
            bid += 1
        if bid > 0:
            i -= self.bins[bid - 1]
        return self.datasets[bid][i]


class DataLoader:
    """DataLoader(shuffleobs(data), batch_size) (src/Monodepth.jl:158): full batches only (partial = false is DataLoaders.jl's
    default), samples collated into ONE reusable pinned host buffer per slot so the H2D copy can be asynchronous."""

    def __init__(self, dataset, batch_size, shuffle=False, seed=0, pin_memory=True, slots=2, shard=(0, 1)):
        """shard = (rank, world): every rank draws the same permutation (same seed) of GLOBAL batches of batch_size * world samples and decodes only its own
        contiguous slice of each (data parallelism: host IO per rank stays constant as the world grows)."""
        self.dataset, self.batch_size, self.shuffle = dataset, batch_size, shuffle
        self.rng = np.random.default_rng(seed)
        self.pin, self.slots, self._bufs = pin_memory, max(1, slots), None
        self.rank, self.world = shard

    def __len__(self):
        return len(self.dataset) // (self.batch_size * self.world)

    def _buffers(self, shape):
        if self._bufs is None or tuple(self._bufs[0].shape) != shape:
            import torch
            pin = self.pin and torch.cuda.is_available()
            self._bufs = [torch.empty(shape, dtype=torch.float32, pin_memory=pin) for _ in range(self.slots)]
        return self._bufs

    def __iter__(self):
        order = self.rng.permutation(len(self.dataset)) if self.shuffle else np.arange(len(self.dataset))
        G = self.batch_size * self.world
        for b in range(len(self)):
            base = b * G + self.rank * self.batch_size
            first = self.dataset[int(order[base])]
            buf = self._buffers((self.batch_size,) + first.shape)[b % self.slots]
            view = buf.numpy()
            view[0] = first
            for j in range(1, self.batch_size):
                view[j] = self.dataset[int(order[base + j])]
            yield buf


def find_static(dataset, alpha, device="cuda"):
    """src/dtk.jl:51-69: file names of the samples whose mean automasking loss exceeds alpha (the non-static ones)."""
    import torch
    from . import monodepth as md
    ssim = md.SSIM()
    non_static = []
    for i in range(len(dataset)):
        x = torch.from_numpy(dataset[i])[None].to(device)
        a = md.automasking_loss(ssim, x, x[:, dataset.target_id - 1], dataset.source_ids)
        if float(a.mean()) > alpha:
            non_static.append(dataset.files[i])
    return non_static


# ---------------------------------------------------------------------------------------------- PNG writers (src/io_utils.jl)
_THERMAL = np.array([[0.016, 0.138, 0.202], [0.262, 0.184, 0.585], [0.710, 0.310, 0.463], [0.984, 0.600, 0.300], [0.910, 0.981, 0.355]], np.float32)


def thermal(v):
    """5-stop approximation of the cmocean `thermal` map; v in [0,1] -> (..., 3) float."""
    v = np.clip(np.asarray(v, np.float32), 0, 1) * (len(_THERMAL) - 1)
    i0 = np.minimum(v.astype(np.int64), len(_THERMAL) - 2)
    f = (v - i0)[..., None]
    return _THERMAL[i0] * (1 - f) + _THERMAL[i0 + 1] * f


def save_disparity(disparity, path):
    """heat map of one (H, W) disparity, colour range = its own min..max (Plots.heatmap default)."""
    d = np.asarray(disparity, np.float32)
    d = d.reshape(d.shape[-2:])
    lo, hi = float(d.min()), float(d.max())
    rgb = thermal((d - lo) / (hi - lo) if hi > lo else np.zeros_like(d))
    Image.fromarray((rgb * 255.0 + 0.5).astype(np.uint8)).save(path)


def save_warped(warped, path):
    """one warped source image, (C, H, W) or (H, W), C = 1 or 3."""
    w = np.clip(np.asarray(warped, np.float32), 0, 1)
    if w.ndim == 3 and w.shape[0] == 1:
        w = w[0]
    if w.ndim == 3:
        w = np.transpose(w, (1, 2, 0))
    Image.fromarray((w * 255.0 + 0.5).astype(np.uint8)).save(path)
