"""ORACLE (test infrastructure, not product code): torch-CPU fp32 restatement of the
reference's RAFT / WeightedRAFT forward pass, written functionally over a state-dict.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; woft_amd/ never does.

Pinned against the imported reference (see oracle/gen_golden.py, tests/golden/*.npz and
tests/test_oracle_golden.py): the reference has no tests of its own,
so the golden vectors are outputs of the reference modules run in the build container.

Citations are relative to /root/reference/pytracking/external/RAFT/raft_core/ .
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------
# encoders  (extractor.py)
# ----------------------------------------------------------------------------------
def _conv(sd, name, x, stride=1, padding=0):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=padding)


def _norm(sd, name, x, kind):
    if kind == "instance":                      # nn.InstanceNorm2d defaults: eps 1e-5, no affine
        return F.instance_norm(x, eps=1e-5)
    if kind == "batch":                         # eval-mode BatchNorm2d
        return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                            sd[name + ".weight"], sd[name + ".bias"], False, 0.0, 1e-5)
    assert kind == "none"
    return x


def residual_block(sd, p, x, kind, stride):
    """extractor.py:48-56 -- note relu is applied to y BEFORE the skip add."""
    y = F.relu(_norm(sd, p + ".norm1", _conv(sd, p + ".conv1", x, stride, 1), kind))
    y = F.relu(_norm(sd, p + ".norm2", _conv(sd, p + ".conv2", y, 1, 1), kind))
    if stride != 1:
        x = _norm(sd, p + ".norm3", _conv(sd, p + ".downsample.0", x, stride, 0), kind)
    return F.relu(x + y)


def bottleneck_block(sd, p, x, kind, stride):
    """extractor.py:107-116."""
    y = F.relu(_norm(sd, p + ".norm1", _conv(sd, p + ".conv1", x, 1, 0), kind))
    y = F.relu(_norm(sd, p + ".norm2", _conv(sd, p + ".conv2", y, stride, 1), kind))
    y = F.relu(_norm(sd, p + ".norm3", _conv(sd, p + ".conv3", y, 1, 0), kind))
    if stride != 1:
        x = _norm(sd, p + ".norm4", _conv(sd, p + ".downsample.0", x, stride, 0), kind)
    return F.relu(x + y)


def encoder(sd, p, x, kind, small):
    """BasicEncoder.forward extractor.py:168-192 / SmallEncoder.forward extractor.py:244-267."""
    block = bottleneck_block if small else residual_block
    x = F.relu(_norm(sd, p + ".norm1", _conv(sd, p + ".conv1", x, 2, 3), kind))
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        x = block(sd, f"{p}.layer{li}.0", x, kind, stride)
        x = block(sd, f"{p}.layer{li}.1", x, kind, 1)
    return _conv(sd, p + ".conv2", x)


# ----------------------------------------------------------------------------------
# correlation volume + lookup  (corr.py, utils/utils.py)
# ----------------------------------------------------------------------------------
def corr_pyramid(fmap1, fmap2, num_levels=4):
    """CorrBlock.__init__ / CorrBlock.corr, corr.py:13-27,62-69.
    Returns a list of (B*H1*W1, 1, H2/2^l, W2/2^l) tensors."""
    b, d, h, w = fmap1.shape
    vol = torch.matmul(fmap1.view(b, d, h * w).transpose(1, 2), fmap2.view(b, d, h * w))
    vol = vol / torch.sqrt(torch.tensor(d).float())
    vol = vol.reshape(b * h * w, 1, h, w)
    pyr = [vol]
    for _ in range(num_levels - 1):
        vol = F.avg_pool2d(vol, 2, stride=2)
        pyr.append(vol)
    return pyr


def bilinear_sampler(img, coords):
    """utils/utils.py:59-73 -- pixel coords -> grid_sample(align_corners=True), zero padding."""
    H, W = img.shape[-2:]
    xg, yg = coords.split([1, 1], dim=-1)
    xg = 2 * xg / (W - 1) - 1
    yg = 2 * yg / (H - 1) - 1
    return F.grid_sample(img, torch.cat([xg, yg], dim=-1), align_corners=True)


def corr_lookup(pyr, coords, radius):
    """CorrBlock.__call__, corr.py:29-59.  coords (B,2,H1,W1) -> (B, L*(2r+1)^2, H1, W1).

    The window offset grid is meshgrid(dy, dx) stacked on the last axis and added to
    (x, y): window element (i, j) samples x + (i - r), y + (j - r)  (x-major window).
    """
    r = radius
    coords = coords.permute(0, 2, 3, 1)
    b, h1, w1, _ = coords.shape
    d = torch.linspace(-r, r, 2 * r + 1)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1).view(1, 2 * r + 1, 2 * r + 1, 2)
    out = []
    for lvl, vol in enumerate(pyr):
        centroid = coords.reshape(b * h1 * w1, 1, 1, 2) / 2 ** lvl
        s = bilinear_sampler(vol, centroid + delta)
        out.append(s.view(b, h1, w1, -1))
    return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous().float()


def lookup_direct(pyr, coords, radius):
    """The same lookup written in pixel coordinates without grid_sample's normalise /
    un-normalise round trip: this is the formula the HIP kernel implements
    (floor, 4 taps, zeros outside).  Differs from corr_lookup by <~5e-5 abs."""
    r = radius
    b, _, h1, w1 = coords.shape
    P = b * h1 * w1
    cx = coords[:, 0].reshape(P)
    cy = coords[:, 1].reshape(P)
    n = 2 * r + 1
    out = []
    for lvl, vol in enumerate(pyr):
        v = vol[:, 0]
        H2, W2 = v.shape[-2:]
        x = cx / 2 ** lvl
        y = cy / 2 ** lvl
        x0 = torch.floor(x)
        y0 = torch.floor(y)
        fx = (x - x0)[:, None, None]
        fy = (y - y0)[:, None, None]
        ii = torch.arange(n)
        # integer patch (n+1)x(n+1): rows = y0-r .. y0+r+1, cols = x0-r .. x0+r+1
        ys = (y0.long()[:, None] - r + torch.arange(n + 1)[None])
        xs = (x0.long()[:, None] - r + torch.arange(n + 1)[None])
        ok = ((ys >= 0) & (ys < H2))[:, :, None] & ((xs >= 0) & (xs < W2))[:, None, :]
        patch = v[torch.arange(P)[:, None, None], ys.clamp(0, H2 - 1)[:, :, None], xs.clamp(0, W2 - 1)[:, None, :]]
        patch = patch * ok
        # window element (i, j): x offset i, y offset j  -> patch[j + dy, i + dx]
        tl = patch[:, :n, :n]
        tr = patch[:, :n, 1:]
        bl = patch[:, 1:, :n]
        br = patch[:, 1:, 1:]
        s = tl * (1 - fx) * (1 - fy) + tr * fx * (1 - fy) + bl * (1 - fx) * fy + br * fx * fy   # [P, j(y), i(x)]
        s = s.permute(0, 2, 1)                     # -> [P, i, j]
        out.append(s.reshape(b, h1, w1, n * n))
        del ii
    return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous()


def coords_grid(b, h, w):
    """utils/utils.py:76-79: channel 0 = x, channel 1 = y."""
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack([xs, ys], dim=0).float()[None].repeat(b, 1, 1, 1)


# ----------------------------------------------------------------------------------
# update block  (update.py)
# ----------------------------------------------------------------------------------
def motion_encoder(sd, p, flow, corr, small):
    """BasicMotionEncoder.forward update.py:89-97 / SmallMotionEncoder.forward update.py:71-77."""
    cor = F.relu(_conv(sd, p + ".convc1", corr))
    if not small:
        cor = F.relu(_conv(sd, p + ".convc2", cor, 1, 1))
    flo = F.relu(_conv(sd, p + ".convf1", flow, 1, 3))
    flo = F.relu(_conv(sd, p + ".convf2", flo, 1, 1))
    out = F.relu(_conv(sd, p + ".conv", torch.cat([cor, flo], dim=1), 1, 1))
    return torch.cat([out, flow], dim=1)


def _gru_half(sd, p, suffix, h, x, pad):
    hx = torch.cat([h, x], dim=1)
    z = torch.sigmoid(_conv(sd, f"{p}.convz{suffix}", hx, 1, pad))
    r = torch.sigmoid(_conv(sd, f"{p}.convr{suffix}", hx, 1, pad))
    q = torch.tanh(_conv(sd, f"{p}.convq{suffix}", torch.cat([r * h, x], dim=1), 1, pad))
    return (1 - z) * h + z * q


def sep_conv_gru(sd, p, h, x):
    """SepConvGRU.forward update.py:45-60: (1x5) pass then (5x1) pass."""
    h = _gru_half(sd, p, "1", h, x, (0, 2))
    return _gru_half(sd, p, "2", h, x, (2, 0))


def conv_gru(sd, p, h, x):
    """ConvGRU.forward update.py:23-31 (small model, 3x3)."""
    return _gru_half(sd, p, "", h, x, 1)


def flow_head(sd, p, x):
    """FlowHead.forward update.py:13-14."""
    return _conv(sd, p + ".conv2", F.relu(_conv(sd, p + ".conv1", x, 1, 1)), 1, 1)


def update_block(sd, net, inp, corr, flow, small):
    """BasicUpdateBlock.forward update.py:127-136 / SmallUpdateBlock.forward update.py:106-112."""
    p = "update_block"
    mf = motion_encoder(sd, p + ".encoder", flow, corr, small)
    x = torch.cat([inp, mf], dim=1)
    if small:
        net = conv_gru(sd, p + ".gru", net, x)
        return net, None, flow_head(sd, p + ".flow_head", net)
    net = sep_conv_gru(sd, p + ".gru", net, x)
    delta = flow_head(sd, p + ".flow_head", net)
    mask = 0.25 * _conv(sd, p + ".mask.2", F.relu(_conv(sd, p + ".mask.0", net, 1, 1)))
    return net, mask, delta


# ----------------------------------------------------------------------------------
# upsampling  (weighted_raft.py:92-103, utils/utils.py:82-84)
# ----------------------------------------------------------------------------------
def convex_upsample(x, mask):
    """out[c, 8h+i, 8w+j] = sum_k softmax_k(mask[k*64+i*8+j, h, w]) * 8*x[c, h+ky-1, w+kx-1],
    k = ky*3+kx, zero padding."""
    n, c, h, w = x.shape
    m = torch.softmax(mask.view(n, 1, 9, 8, 8, h, w), dim=2)
    u = F.unfold(8 * x, [3, 3], padding=1).view(n, c, 9, 1, 1, h, w)
    u = torch.sum(m * u, dim=2)
    return u.permute(0, 1, 4, 2, 5, 3).reshape(n, c, 8 * h, 8 * w)


def upflow8(x):
    return 8 * F.interpolate(x, size=(8 * x.shape[2], 8 * x.shape[3]), mode="bilinear", align_corners=True)


# ----------------------------------------------------------------------------------
# weight head  (weighted_raft.py:258-279, 347-384)
# ----------------------------------------------------------------------------------
def weight_head(sd, lookup, vol0, radius, hf, wf):
    """lookup: (B, L*n*n, H1, W1) final lookup; vol0: (B*H1*W1, 1, H2, W2) level-0 volume.

    The reference re-reads the lookup channels as (H_patch W_patch N_levels)
    (weighted_raft.py:267-272) although they are laid out level-major: input channel
    n, window position (hp, wp) of the head is lookup channel hp*(n_w*L) + wp*L + n.
    """
    b = lookup.shape[0]
    n = 2 * radius + 1
    L = 4
    samp = lookup.view(b, n, n, L, hf, wf)                       # (B hp wp lvl H1 W1)
    mean = vol0.view(b, hf, wf, -1).mean(dim=-1)                 # (B H1 W1)
    x = samp.permute(0, 4, 5, 3, 1, 2).reshape(b * hf * wf, L, n, n)
    m = mean.reshape(b * hf * wf, 1, 1, 1).expand(-1, 1, n, n)
    x = torch.cat([x, m], dim=1)
    # any weight_head_structure (weighted_raft.py:318-345): convs net.0, net.2, ... with padding k // 2 and a ReLU each, then
    # the closing 1x1 conv (the shipped configs: three 3x3 layers of 128 channels)
    p = "weight_head.net"
    idx = sorted(int(k.split(".")[2]) for k in sd if k.startswith(p + ".") and k.endswith(".weight"))
    for i in idx[:-1]:
        k = sd[f"{p}.{i}.weight"].shape[-1]
        x = F.relu(_conv(sd, f"{p}.{i}", x, 1, k // 2))
    x = _conv(sd, f"{p}.{idx[-1]}", x)
    return x.view(b, hf, wf, n * n).mean(dim=-1)[:, None]


# ----------------------------------------------------------------------------------
# full forward passes
# ----------------------------------------------------------------------------------
def raft_forward(sd, image1, image2, iters, small=False, weighted=True, trace=None):
    """WeightedRAFT.forward (weighted_raft.py:179-315) / RAFT.forward (raft.py:169-262),
    test_mode=True, mixed_precision=False, alternate_corr=False, flow_init=None.

    image1/2: (1,3,H,W) float RGB in [0,255].  Returns dict with flow_low, flow_up and,
    when weighted, weights_low (logits at 1/8), weights_up (logits at full res).
    `trace`, if a dict, receives intermediates for kernel-level tests.
    """
    hdim, cdim, radius = (96, 64, 3) if small else (128, 128, 4)
    image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
    image2 = (2 * (image2 / 255.0) - 1.0).contiguous()

    f = encoder(sd, "fnet", torch.cat([image1, image2], 0), "instance", small)
    fmap1, fmap2 = f[:1].float(), f[1:].float()
    pyr = corr_pyramid(fmap1, fmap2)

    c = encoder(sd, "cnet", image1, "none" if small else "batch", small)
    net, inp = torch.split(c, [hdim, cdim], dim=1)
    net, inp = torch.tanh(net), torch.relu(inp)

    b, _, H, W = image1.shape
    coords0 = coords_grid(b, H // 8, W // 8)
    coords1 = coords0.clone()
    if trace is not None:
        trace.update(fmap1=fmap1, fmap2=fmap2, pyr=pyr, net0=net, inp=inp, lookups=[], nets=[], coords=[])
    up_mask = None
    for _ in range(iters):
        corr = corr_lookup(pyr, coords1, radius)
        net, up_mask, delta = update_block(sd, net, inp, corr, coords1 - coords0, small)
        coords1 = coords1 + delta
        if trace is not None:
            trace["lookups"].append(corr)
            trace["nets"].append(net)
            trace["coords"].append(coords1)
    flow = coords1 - coords0
    flow_up = upflow8(flow) if up_mask is None else convex_upsample(flow, up_mask)
    out = dict(flow_low=flow, flow_up=flow_up)
    if trace is not None:
        trace["up_mask"] = up_mask
    if weighted:
        corr = corr_lookup(pyr, coords1, radius)
        w = weight_head(sd, corr, pyr[0], radius, H // 8, W // 8)
        w_up = (upflow8(w) if up_mask is None else convex_upsample(w, up_mask)) / 8
        out.update(weights_low=w, weights_up=w_up)
        if trace is not None:
            trace["final_lookup"] = corr
    return out


# ----------------------------------------------------------------------------------
# operator boundary: RAFTWrapper.compute_flow  (pytracking/optical_flow/raft.py:81-218)
# ----------------------------------------------------------------------------------
def pad_inputs(src, dst, mode):
    """Padding policies optical_flow/raft.py:122-132,221-271 and utils/utils.py:7-26.
    Returns (src, dst, unpad_fn)."""
    _, _, H, W = dst.shape
    if mode == "nopad":
        assert H % 8 == 0 and W % 8 == 0
        return src, dst, (lambda t: t)
    if mode == "crop":
        ch, cw = (H // 8) * 8, (W // 8) * 8
        return src[:, :, :ch, :cw], dst[:, :, :ch, :cw], (lambda t: t)
    if mode == "RAFT":
        ph = (((H // 8) + 1) * 8 - H) % 8
        pw = (((W // 8) + 1) * 8 - W) % 8
        pad = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]

        def unpad(t):
            if t is None:
                return None
            hh, ww = t.shape[-2:]
            return t[..., pad[2]:hh - pad[3], pad[0]:ww - pad[1]]
        return F.pad(src, pad, mode="replicate"), F.pad(dst, pad, mode="replicate"), unpad
    if mode == "Michal":
        hn, wn = int(math.ceil(H / 8) * 8), int(math.ceil(W / 8) * 8)

        def unpad(t):
            if t is None:
                return None
            r = F.interpolate(t, size=(H, W), mode="bilinear")
            if t.shape[1] == 2:
                return torch.cat([r[:, 0:1] * W / wn, r[:, 1:2] * H / hn], dim=1)
            # the reference's MichalPadder.unpad on a 1-channel tensor concatenates
            # r[:,0:1]*W/wn with an empty slice (optical_flow/raft.py:262-271)
            return r[:, 0:1] * W / wn
        return (F.interpolate(src, size=(hn, wn), mode="bilinear"),
                F.interpolate(dst, size=(hn, wn), mode="bilinear"), unpad)
    raise ValueError(f"invalid padding_mode '{mode}'")


def compute_flow(sd, src_bgr, dst_bgr, iters, mode="TC", small=False, weighted=True,
                 padding_mode="nopad", do_sigmoid=False):
    """numpy uint8 BGR (H,W,3) pair -> the tuples RAFTWrapper.compute_flow returns (on CPU)."""
    assert mode in ("flow", "TC")
    assert src_bgr.shape == dst_bgr.shape
    to_t = lambda a: torch.from_numpy(a[:, :, ::-1].copy()).permute(2, 0, 1).float()[None]
    src, dst, unpad = pad_inputs(to_t(src_bgr), to_t(dst_bgr), padding_mode)
    out = raft_forward(sd, src, dst, iters, small=small, weighted=weighted)
    flow = unpad(out["flow_up"])
    w = unpad(out["weights_up"]) if weighted else None
    if do_sigmoid and w is not None:
        w = torch.sigmoid(w)
    if mode == "flow":
        return flow[0], (w[0] if w is not None else None)
    _, _, H, W = flow.shape
    idx = torch.arange(H * W)
    src_coords = torch.stack([idx % W, torch.div(idx, W, rounding_mode="floor")], dim=0)   # misc.py:45-96
    dst_coords = src_coords + flow.reshape(2, H * W)
    return src_coords, dst_coords, (w.reshape(1, H * W) if w is not None else None)
