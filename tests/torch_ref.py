"""Independent torch-CPU implementation of the documented Flux/NNlib op semantics.

NOT the reference and NOT the oracle: a second, library-based implementation used (a) in the build
container to cross-check the C++ oracle (forward and autograd gradients) and (b) to generate the
small committed fixtures under tests/golden/ ("torch-derived, not Flux-derived").  Layout NCHW ==
Julia WHCN bytes.
"""
import math
import torch
import torch.nn.functional as F


# ---------------- L2 ops (src/utils.jl) ----------------
def hat(r):  # r (N,3) -> (N,3,3)
    N = r.shape[0]; S = torch.zeros(N, 3, 3, dtype=r.dtype)
    S[:, 1, 0] = r[:, 2]; S[:, 0, 1] = -r[:, 2]; S[:, 2, 0] = -r[:, 1]; S[:, 0, 2] = r[:, 1]; S[:, 2, 1] = r[:, 0]; S[:, 1, 2] = -r[:, 0]
    return S


def so3_exp_map(r):
    S = hat(r); S2 = S @ S
    th = torch.sqrt((r * r).sum(1))
    thi = 1.0 / torch.clamp(th, min=1e-4)
    f1 = (thi * torch.sin(th)).view(-1, 1, 1); f2 = (thi * thi * (1 - torch.cos(th))).view(-1, 1, 1)
    return f1 * S + f2 * S2 + torch.eye(3, dtype=r.dtype)


def composeT(r, t, invert):  # t (N,3)
    R = so3_exp_map(r)
    if invert:
        R = R.transpose(1, 2); t = (R @ (-t).unsqueeze(-1)).squeeze(-1)
    return R, t


def ssim(x, y):
    xr = F.pad(x, (1, 1, 1, 1), mode="reflect"); yr = F.pad(y, (1, 1, 1, 1), mode="reflect")
    pool = lambda a: F.avg_pool2d(a, 3, 1)
    mx, my = pool(xr), pool(yr)
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    sx = pool(xr * xr) - mx * mx; sy = pool(yr * yr) - my * my; sxy = pool(xr * yr) - mx * my
    n = (2 * mx * my + c1) * (2 * sxy + c2); d = (mx * mx + my * my + c1) * (sx + sy + c2)
    return torch.clamp((1 - n / d) * 0.5, 0, 1)


def photometric_loss(pred, tgt, alpha=0.85):
    l1 = (tgt - pred).abs().mean(1, keepdim=True); s = ssim(pred, tgt).mean(1, keepdim=True)
    return alpha * s + (1 - alpha) * l1


def prediction_loss(preds, tgt):
    return torch.cat([photometric_loss(p, tgt) for p in preds], 1).min(1, keepdim=True)[0]


def disparity_to_depth(d, mn, mx):
    return 1.0 / (d * (1.0 / mn - 1.0 / mx) + 1.0 / mx)


def backproject(depth, invK, H, W):  # depth (N,P) -> (N,P,3)
    ws = torch.arange(1, W + 1, dtype=depth.dtype); hs = torch.arange(1, H + 1, dtype=depth.dtype)
    hh, ww = torch.meshgrid(hs, ws, indexing="ij")
    coords = torch.stack([ww.reshape(-1), hh.reshape(-1), torch.ones(H * W, dtype=depth.dtype)], 0)  # (3,P)
    rays = (invK @ coords).t()  # (P,3)
    return depth.unsqueeze(-1) * rays.unsqueeze(0)


def project(pts, K, R, t, H, W):  # pts (N,P,3), R (N,3,3), t (N,3) -> (N,P,2)
    cam = (pts @ R.transpose(1, 2) + t.unsqueeze(1)) @ K.t()
    pix = cam[..., :2] / (cam[..., 2:3] + 1e-7)
    norm = torch.tensor([W - 1.0, H - 1.0], dtype=pts.dtype)
    return ((pix - 1) / norm - 0.5) * 2


def smooth_loss(d, img):  # d (N,H,W), img (N,C,H,W)
    dx = (d[:, :, :-1] - d[:, :, 1:]).abs(); dy = (d[:, :-1, :] - d[:, 1:, :]).abs()
    ix = (img[:, :, :, :-1] - img[:, :, :, 1:]).abs().mean(1); iy = (img[:, :, :-1, :] - img[:, :, 1:, :]).abs().mean(1)
    return (dx * torch.exp(-ix)).mean() + (dy * torch.exp(-iy)).mean()


def grid_sample(inp, grid, border=True):
    return F.grid_sample(inp, grid, mode="bilinear", padding_mode="border" if border else "zeros", align_corners=True)


def up_to(x, H, W):
    return F.interpolate(x, size=(H, W), mode="bilinear", align_corners=True)


# ---------------- layers (Flux semantics) ----------------
def flux_conv(x, w, b, stride=1, pad=0):
    """w in Flux layout (Cout,Cin,KH,KW) == (kW,kH,Cin,Cout) col-major; TRUE convolution -> flip spatially."""
    return F.conv2d(x, torch.flip(w, (2, 3)), b, stride=stride, padding=pad)


class BNState:
    """Tracked BatchNorm statistics in call order (torch's running_mean / running_var use Flux's convention: momentum 0.1 on the
    new value, unbiased batch variance), and the train / test switch."""

    def __init__(self):
        self.stats, self.i, self.test = [], 0, False

    def flat(self):
        return torch.cat([torch.cat([m, v]) for m, v in self.stats]).numpy()


_BN = None          # set by eval_disparity(..., bn=BNState) for the duration of one forward pass


def bn_train(x, g, b):
    if _BN is None:
        return F.batch_norm(x, None, None, g, b, training=True, momentum=0.1, eps=1e-5)
    i = _BN.i; _BN.i += 1
    if i >= len(_BN.stats):
        _BN.stats.append((torch.zeros(x.shape[1], dtype=x.dtype), torch.ones(x.shape[1], dtype=x.dtype)))
    rm, rv = _BN.stats[i]
    return F.batch_norm(x, rm, rv, g, b, training=not _BN.test, momentum=0.1, eps=1e-5)


class Params:
    """Walks the flat parameter vector in the documented order."""

    def __init__(self, flat):
        self.flat = flat; self.o = 0

    def conv(self, cin, cout, k, bias):
        n = cin * cout * k * k; w = self.flat[self.o:self.o + n].view(cout, cin, k, k); self.o += n
        b = None
        if bias:
            b = self.flat[self.o:self.o + cout]; self.o += cout
        return w, b

    def bn(self, c):
        g = self.flat[self.o:self.o + c]; b = self.flat[self.o + c:self.o + 2 * c]; self.o += 2 * c
        return g, b


def model_forward(flat, x, C, target_id=2, source_ids=(1, 3), depth=18, scale_levels=(2, 3, 4, 5)):
    """x (N,L,C,H,W) -> (disps (N,1,h,w) per scale level, [(rvec (N,3), tvec (N,3))]*2).  src/model.jl:8-20"""
    P = Params(flat); N, L = x.shape[:2]
    h = x.reshape(N * L, C, x.shape[3], x.shape[4])
    # encoder (torchvision ResNet-18 / ResNet-50 trunk)
    w, _ = P.conv(C, 64, 7, False); g, b = P.bn(64)
    h = F.relu(bn_train(flux_conv(h, w, None, 2, 3), g, b)); feats = [h]
    h = F.max_pool2d(h, 3, 2, 1)
    cin = 64
    nblocks = {18: (2, 2, 2, 2), 50: (3, 4, 6, 3)}[depth]
    for li, planes in enumerate((64, 128, 256, 512)):
        for bi in range(nblocks[li]):
            s = 2 if (bi == 0 and li > 0) else 1
            if depth == 18:
                co = planes
                w1, _ = P.conv(cin, co, 3, False); g1, b1 = P.bn(co); w2, _ = P.conv(co, co, 3, False); g2, b2 = P.bn(co)
                o = F.relu(bn_train(flux_conv(h, w1, None, s, 1), g1, b1)); o = bn_train(flux_conv(o, w2, None, 1, 1), g2, b2)
            else:                                  # bottleneck, stride on the 3x3
                co = 4 * planes
                w1, _ = P.conv(cin, planes, 1, False); g1, b1 = P.bn(planes); w2, _ = P.conv(planes, planes, 3, False); g2, b2 = P.bn(planes)
                w3, _ = P.conv(planes, co, 1, False); g3, b3 = P.bn(co)
                o = F.relu(bn_train(flux_conv(h, w1, None, 1, 0), g1, b1)); o = F.relu(bn_train(flux_conv(o, w2, None, s, 1), g2, b2))
                o = bn_train(flux_conv(o, w3, None, 1, 0), g3, b3)
            if s != 1 or cin != co:
                wd, _ = P.conv(cin, co, 1, False); gd, bd = P.bn(co); sc = bn_train(flux_conv(h, wd, None, s, 0), gd, bd)
            else:
                sc = h
            h = F.relu(o + sc); cin = co
        feats.append(h)
    f5 = [f.view(N, L, *f.shape[1:]) for f in feats]
    tf = [f[:, target_id - 1] for f in f5]
    # depth decoder
    dec = [256, 128, 64, 32, 16]; inc = [cin, 256, 128, 64, 32]; skc = [feats[3].shape[1], feats[2].shape[1], feats[1].shape[1], 64, 0]
    nbr = max(scale_levels)                      # DepthDecoder: branches 1..max(scale_levels), one head per listed level (src/depth_decoder.jl:38-46)
    dconvs = [(P.conv(inc[b_], dec[b_], 3, True), P.conv(dec[b_] + skc[b_], dec[b_], 3, True)) for b_ in range(nbr)]
    heads = {l: P.conv(dec[l - 1], 1, 3, True) for l in scale_levels}
    rp = lambda a: F.pad(a, (1, 1, 1, 1), mode="reflect")
    xx = tf[4]; disps = []
    for b_ in range(nbr):
        (w1, b1), (w2, b2) = dconvs[b_]
        xx = F.elu(flux_conv(rp(xx), w1, b1)); xx = F.interpolate(xx, scale_factor=2, mode="bilinear", align_corners=True)
        if b_ < 4:
            xx = torch.cat([xx, tf[3 - b_]], 1)
        xx = F.elu(flux_conv(rp(xx), w2, b2))
        if b_ + 1 in heads:
            wh, bh = heads[b_ + 1]; disps.append(torch.sigmoid(flux_conv(rp(xx), wh, bh)))
    # pose decoder
    ws, bs = P.conv(cin, 256, 1, True); wp1, bp1 = P.conv(512, 256, 3, True); wp2, bp2 = P.conv(256, 256, 3, True); wp3, bp3 = P.conv(256, 6, 1, True)
    assert P.o == flat.numel(), (P.o, flat.numel())
    poses = []
    for sid in source_ids:
        pair = (f5[4][:, sid - 1], tf[4]) if sid < target_id else (tf[4], f5[4][:, sid - 1])
        sq = torch.cat([F.relu(flux_conv(p, ws, bs)) for p in pair], 1)
        y = F.relu(flux_conv(sq, wp1, bp1, 1, 1)); y = F.relu(flux_conv(y, wp2, bp2, 1, 1)); y = flux_conv(y, wp3, bp3)
        pose = 0.01 * y.mean((2, 3))
        poses.append((pose[:, :3], pose[:, 3:]))
    return disps, poses


def loss_from_outputs(x, disps, poses, K, invK, target_id=2, source_ids=(1, 3), min_depth=0.1, max_depth=100.0,
                      smoothness=1e-3, scales=(0.125, 0.25, 0.5, 1.0), auto_loss=None):
    """The loss half of train_loss, src/training.jl:28-77."""
    N, L, C, H, W = x.shape
    tgt = x[:, target_id - 1]
    Ps = [composeT(r, t, sid < target_id) for (r, t), sid in zip(poses, source_ids)]
    loss = 0.0
    for d, sc in zip(disps, scales):
        if d.shape[2] != H or d.shape[3] != W:
            d = up_to(d, H, W)
        depth = disparity_to_depth(d, min_depth, max_depth)
        coords = backproject(depth.reshape(N, H * W), invK, H, W)
        warped = []
        for (R, t), sid in zip(Ps, source_ids):
            uv = project(coords, K, R, t, H, W).view(N, H, W, 2)
            warped.append(grid_sample(x[:, sid - 1], uv, True))
        wl = prediction_loss(warped, tgt)
        if auto_loss is not None:
            wl = torch.cat([auto_loss, wl], 1).min(1, keepdim=True)[0]
        nd = (d / (d.mean((2, 3), keepdim=True) + 1e-7))[:, 0]
        loss = loss + wl.mean() + smooth_loss(nd, tgt) * smoothness * sc
    return loss / len(scales)


def train_loss(flat, x, C, K, invK, **kw):
    disps, poses = model_forward(flat, x, C)
    return loss_from_outputs(x, disps, poses, K, invK, **kw)


def automasking_loss(x, target_id=2, source_ids=(1, 3)):
    tgt = x[:, target_id - 1]
    return torch.cat([photometric_loss(x[:, s - 1], tgt) for s in source_ids], 1).min(1, keepdim=True)[0]


def adam_step(p, g, m, v, t, eta=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    m.mul_(b1).add_(g, alpha=1 - b1); v.mul_(b2).addcmul_(g, g, value=1 - b2)
    p.sub_(m / (1 - b1 ** t) / (torch.sqrt(v / (1 - b2 ** t)) + eps) * eta)


def simple_depth_loss(x, disp, poses, K, invK, target_id=2, source_ids=(1, 3), min_depth=0.1, max_depth=100.0):
    """slow_depth's objective, src/simple_depth.jl:25-42: mean(prediction_loss(warp(...))) + smooth_loss(disp) (weight 1, raw disparity)."""
    N, L, C, H, W = x.shape
    tgt = x[:, target_id - 1]
    Ps = [composeT(r, t, sid < target_id) for (r, t), sid in zip(poses, source_ids)]
    depth = disparity_to_depth(disp, min_depth, max_depth)
    coords = backproject(depth.reshape(N, H * W), invK, H, W)
    warped = [grid_sample(x[:, sid - 1], project(coords, K, R, t, H, W).view(N, H, W, 2), True) for (R, t), sid in zip(Ps, source_ids)]
    return prediction_loss(warped, tgt).mean() + smooth_loss(disp[:, 0], tgt)


def eval_disparity(flat, x, C, bn=None):
    """eval_disparity (src/model.jl:28): encoder + depth decoder on single frames x (N,C,H,W); BN on batch statistics, or with a BNState:
    tracked statistics updated (train mode) / used (bn.test, Flux.testmode!)."""
    global _BN
    _BN = bn
    if bn is not None:
        bn.i = 0
    try:
        disps, _ = model_forward(flat, x.unsqueeze(1), C, target_id=1, source_ids=())
    finally:
        _BN = None
    return disps
