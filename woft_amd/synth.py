"""Deterministic synthetic checkpoints and image sequences.

The reference's trained checkpoints are absent from the snapshot
(/root/reference/.MISSING_LARGE_BLOBS), so every parity test, the smoke test and
the benchmark use state-dicts generated here from a seed.  The key set and the
tensor shapes follow the reference modules:

  * encoders            pytracking/external/RAFT/raft_core/extractor.py:118-267
  * update blocks       pytracking/external/RAFT/raft_core/update.py:6-136
  * weight head         pytracking/external/RAFT/raft_core/weighted_raft.py:318-345

(the key list is pinned against the imported reference in
tests/golden/state_dict_keys.json).  Everything is numpy.random.RandomState, so the
same seed gives the same bytes everywhere (this container, the GPU box).

The synthetic video (template + homography chain) follows SURVEY.md section 8(d).
"""
from collections import OrderedDict

import numpy as np
import torch


# ----------------------------------------------------------------------------
# state dicts
# ----------------------------------------------------------------------------
class _Gen:
    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.sd = OrderedDict()

    def conv(self, name, cout, cin, kh, kw, mode="kaiming_out", gain=1.0):
        if mode == "kaiming_out":      # extractor.py:150-152 (fan_out, relu)
            std = np.sqrt(2.0 / (cout * kh * kw))
            w = self.rs.standard_normal((cout, cin, kh, kw)) * std
        else:                          # torch default: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            b = 1.0 / np.sqrt(cin * kh * kw)
            w = self.rs.uniform(-b, b, (cout, cin, kh, kw))
        bb = 1.0 / np.sqrt(cin * kh * kw)
        bias = self.rs.uniform(-bb, bb, (cout,))
        self.sd[name + ".weight"] = torch.from_numpy((gain * w).astype(np.float32))
        self.sd[name + ".bias"] = torch.from_numpy((gain * bias).astype(np.float32))

    def bn(self, name, c):
        self.sd[name + ".weight"] = torch.from_numpy(self.rs.uniform(0.8, 1.2, c).astype(np.float32))
        self.sd[name + ".bias"] = torch.from_numpy(self.rs.uniform(-0.1, 0.1, c).astype(np.float32))
        self.sd[name + ".running_mean"] = torch.from_numpy(self.rs.uniform(-0.1, 0.1, c).astype(np.float32))
        self.sd[name + ".running_var"] = torch.from_numpy(self.rs.uniform(0.7, 1.3, c).astype(np.float32))
        self.sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)

    def alias(self, dst, src):
        for k in list(self.sd.keys()):
            if k.startswith(src + "."):
                self.sd[dst + k[len(src):]] = self.sd[k].clone()


def _basic_encoder(g, p, out_dim, norm):
    def nrm(name, c):
        if norm == "batch":
            g.bn(name, c)
    g.conv(p + ".conv1", 64, 3, 7, 7)
    nrm(p + ".norm1", 64)
    cin = 64
    for li, (dim, stride) in enumerate([(64, 1), (96, 2), (128, 2)], start=1):
        for bi in range(2):
            q = f"{p}.layer{li}.{bi}"
            s = stride if bi == 0 else 1
            g.conv(q + ".conv1", dim, cin, 3, 3)
            g.conv(q + ".conv2", dim, dim, 3, 3)
            nrm(q + ".norm1", dim)
            nrm(q + ".norm2", dim)
            if s != 1:
                nrm(q + ".norm3", dim)
                g.conv(q + ".downsample.0", dim, cin, 1, 1)
                if norm == "batch":
                    # norm3 and downsample.1 are the same module in the reference
                    # (extractor.py:25-26,44-45): two key prefixes, one set of tensors.
                    g.alias(q + ".downsample.1", q + ".norm3")
            cin = dim
    g.conv(p + ".conv2", out_dim, 128, 1, 1)


def _small_encoder(g, p, out_dim, norm):
    assert norm in ("instance", "none")
    g.conv(p + ".conv1", 32, 3, 7, 7)
    cin = 32
    for li, (dim, stride) in enumerate([(32, 1), (64, 2), (96, 2)], start=1):
        for bi in range(2):
            q = f"{p}.layer{li}.{bi}"
            s = stride if bi == 0 else 1
            g.conv(q + ".conv1", dim // 4, cin, 1, 1)
            g.conv(q + ".conv2", dim // 4, dim // 4, 3, 3)
            g.conv(q + ".conv3", dim, dim // 4, 1, 1)
            if s != 1:
                g.conv(q + ".downsample.0", dim, cin, 1, 1)
            cin = dim
    g.conv(p + ".conv2", out_dim, 96, 1, 1)


def make_state_dict(seed=0, small=False, weighted=True, head_gain=1.0, weight_head_structure=None):
    """Synthetic checkpoint with the reference's key set.
    weight_head_structure: the flow config's `class_params.weight_head_structure` (weighted_raft.py:318-345: a list of
    (channels, kernel) tuples or plain channel counts = 3x3 layers); default [(128, 3)] * 3, the shipped configs' head.

    small=False, weighted=True  -> WeightedRAFT full (weighted_raft.py:62-72)
    small=True,  weighted=False -> plain RAFT-small  (raft.py:49-56)
    """
    g = _Gen(seed)
    if small:
        _small_encoder(g, "fnet", 128, "instance")
        _small_encoder(g, "cnet", 96 + 64, "none")
        u = "update_block"
        g.conv(u + ".encoder.convc1", 96, 4 * 49, 1, 1, "default")
        g.conv(u + ".encoder.convf1", 64, 2, 7, 7, "default")
        g.conv(u + ".encoder.convf2", 32, 64, 3, 3, "default")
        g.conv(u + ".encoder.conv", 80, 128, 3, 3, "default")
        for n in ("convz", "convr", "convq"):
            g.conv(f"{u}.gru.{n}", 96, 96 + 82 + 64, 3, 3, "default")
        g.conv(u + ".flow_head.conv1", 128, 96, 3, 3, "default")
        g.conv(u + ".flow_head.conv2", 2, 128, 3, 3, "default", gain=head_gain)
    else:
        _basic_encoder(g, "fnet", 256, "instance")
        _basic_encoder(g, "cnet", 256, "batch")
        u = "update_block"
        g.conv(u + ".encoder.convc1", 256, 4 * 81, 1, 1, "default")
        g.conv(u + ".encoder.convc2", 192, 256, 3, 3, "default")
        g.conv(u + ".encoder.convf1", 128, 2, 7, 7, "default")
        g.conv(u + ".encoder.convf2", 64, 128, 3, 3, "default")
        g.conv(u + ".encoder.conv", 126, 256, 3, 3, "default")
        for n, (kh, kw) in (("1", (1, 5)), ("2", (5, 1))):
            for gate in ("convz", "convr", "convq"):
                g.conv(f"{u}.gru.{gate}{n}", 128, 384, kh, kw, "default")
        g.conv(u + ".flow_head.conv1", 256, 128, 3, 3, "default")
        g.conv(u + ".flow_head.conv2", 2, 256, 3, 3, "default", gain=head_gain)
        g.conv(u + ".mask.0", 256, 128, 3, 3, "default")
        g.conv(u + ".mask.2", 576, 256, 1, 1, "default")
    if weighted:
        # weight_head_structure [(128,3)]*3 (optical_flow/configs/v2_SNOB_large_g05_RAFT.py:16) unless told otherwise;
        # layer i is net.(2 i) (a ReLU between two convs), the closing 1x1 conv follows (weighted_raft.py:323-341)
        cur = 5
        structure = weight_head_structure if weight_head_structure is not None else [(128, 3)] * 3
        for i, data in enumerate(structure):
            c, k = data if isinstance(data, (list, tuple)) else (data, 3)
            g.conv(f"weight_head.net.{2 * i}", c, cur, k, k, "default")
            cur = c
        g.conv(f"weight_head.net.{2 * len(structure)}", 1, cur, 1, 1, "default")
    return g.sd


# ----------------------------------------------------------------------------
# synthetic frames
# ----------------------------------------------------------------------------
def _bicubic_up(u, H, W):
    t = torch.from_numpy(u)[None, None]
    return torch.nn.functional.interpolate(t, size=(H, W), mode="bicubic", align_corners=False)[0, 0].numpy()


def make_template(H, W, seq_id=0):
    """(H,W,3) uint8 BGR texture: 0.7 * smooth + 0.3 * white noise."""
    rs = np.random.RandomState(1000 + seq_id)
    u1 = rs.uniform(0, 1, (3, H // 8, W // 8)).astype(np.float32)
    u2 = rs.uniform(0, 1, (3, H, W)).astype(np.float32)
    img = np.stack([_bicubic_up(u1[c], H, W) for c in range(3)], 0)
    img = np.clip(0.7 * img + 0.3 * u2, 0, 1)
    return np.ascontiguousarray(np.round(255 * img).astype(np.uint8).transpose(1, 2, 0))


def seq_homography(t, H, W):
    """H_t: template -> frame t (translation, small rotation about centre, tiny perspective)."""
    cx, cy = W / 2.0, H / 2.0
    a = np.deg2rad(0.2 * t)
    ca, sa = np.cos(a), np.sin(a)
    R = np.array([[ca, -sa, cx - ca * cx + sa * cy],
                  [sa, ca, cy - sa * cx - ca * cy],
                  [0, 0, 1.0]])
    T = np.array([[1, 0, 3.0 * t], [0, 1, -2.0 * t], [0, 0, 1.0]])
    Pp = np.array([[1, 0, 0], [0, 1, 0], [1e-6 * t, 0, 1.0]])
    Ht = T @ R @ Pp
    return Ht / Ht[2, 2]


def warp_image_np(img, Hmat, out_hw=None):
    """Float bilinear perspective warp on the CPU: dst(x) = src(H^-1 x), zeros outside.

    Same sampling rule as the HIP warp kernel (csrc/warp.hip) and cv2.warpPerspective's
    geometric convention (tracker/YAOF_tracker_single_control.py:89-91); used to
    synthesise frames, not as a parity reference for OpenCV's fixed-point arithmetic.
    """
    Hh, Ww = img.shape[:2] if out_hw is None else out_hw
    Hi = np.linalg.inv(Hmat)
    ys, xs = np.mgrid[0:Hh, 0:Ww].astype(np.float64)
    d = Hi[2, 0] * xs + Hi[2, 1] * ys + Hi[2, 2]
    sx = (Hi[0, 0] * xs + Hi[0, 1] * ys + Hi[0, 2]) / d
    sy = (Hi[1, 0] * xs + Hi[1, 1] * ys + Hi[1, 2]) / d
    x0 = np.floor(sx).astype(np.int64)
    y0 = np.floor(sy).astype(np.int64)
    fx = (sx - x0)[..., None]
    fy = (sy - y0)[..., None]
    src = img.astype(np.float64)
    if src.ndim == 2:
        src = src[..., None]
    sh, sw = src.shape[:2]

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < sh) & (xx >= 0) & (xx < sw)
        v = src[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)]
        return v * ok[..., None]

    out = (tap(y0, x0) * (1 - fx) * (1 - fy) + tap(y0, x0 + 1) * fx * (1 - fy)
           + tap(y0 + 1, x0) * (1 - fx) * fy + tap(y0 + 1, x0 + 1) * fx * fy)
    if img.ndim == 2:
        out = out[..., 0]
    return out


def make_frame(template, t):
    """Frame t of the synthetic sequence: the template seen through H_t."""
    H, W = template.shape[:2]
    out = warp_image_np(template, seq_homography(t, H, W))
    return np.ascontiguousarray(np.clip(np.round(out), 0, 255).astype(np.uint8))


def make_init_mask(H, W):
    """Centred half-size rectangle, uint8 0/255."""
    m = np.zeros((H, W), np.uint8)
    m[H // 4:3 * H // 4, W // 4:3 * W // 4] = 255
    return m
