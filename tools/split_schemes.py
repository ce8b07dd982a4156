"""Numerics of operand-split schemes for fp32-emulating products on the gfx950 matrix cores (CPU emulation, numpy; DESIGN section 7,
item 0b): relative error of K-term dot products <a, w> against fp64, for

  bf16        : bf16(a) * bf16(w)                                                     1 bf16 MFMA pass per product
  fp16        : fp16(a) * fp16(w)                                                     1
  bf16x3      : hi*hi + hi*lo + lo*hi, hi = bf16(x), lo = bf16(x - hi)                3   (the shipped arithmetic)
  fp16+mx8x2  : fp16(a)*fp16(w) + mx(la)*mx(w) + mx(a)*mx(lw), la = a - fp16(a), ...  1 + 2 * 0.5  (MX-fp8 e4m3 at twice the bf16 rate)
  fp16+mx6x2  : the same with MX-fp6 e2m3 cross terms                                 1 + 2 * 0.27 (MX-fp6 at ~3.75x the bf16 rate)
  fp16+mx4x2  : the same with MX-fp4 e2m1 cross terms                                 1 + 2 * 0.27

mx(x): block-scaled micro-format of v_mfma_scale_f32_32x32x64_f8f6f4 -- 32 consecutive K elements share a power-of-two scale (E8M0)
chosen from the block's largest magnitude, elements rounded to the small float format.  Rates: MI355X_MICROARCH.md / the guide's
micro-benchmarks (bf16 2382, fp16 2178, MX-fp8 4686, MX-fp6 8939, MX-fp4 9099 TFLOP/s).

python tools/split_schemes.py [K] [rows]"""
import sys

import numpy as np


def to_bf16(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def small_float(x, ebits, mbits, emax_val):
    """Round to a tiny float format (sign, ebits exponent bits with subnormals, mbits mantissa bits, largest finite emax_val);
    round to nearest even on the grid, saturating."""
    x = np.asarray(x, np.float64)
    bias = (1 << (ebits - 1)) - 1
    emin = 1 - bias
    mag = np.abs(x)
    e = np.floor(np.log2(np.maximum(mag, 1e-300)))
    e = np.maximum(e, emin)
    q = np.ldexp(1.0, (e - mbits).astype(int))
    r = np.round(mag / q) * q
    return np.sign(x) * np.minimum(r, emax_val)


FMT = {"mx8": (4, 3, 448.0), "mx6": (2, 3, 7.5), "mx4": (2, 1, 6.0)}


def mx(x, fmt):
    """Block-scaled quantisation along the last axis, 32 elements per block."""
    ebits, mbits, vmax = FMT[fmt]
    shp = x.shape
    b = x.reshape(-1, 32).astype(np.float64)
    amax = np.abs(b).max(1, keepdims=True)
    emax_elem = np.floor(np.log2(vmax))
    scale = np.ldexp(1.0, (np.floor(np.log2(np.maximum(amax, 1e-300))) - emax_elem).astype(int))
    return (small_float(b / scale, ebits, mbits, vmax) * scale).reshape(shp)


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 2304          # 3x3 taps x 256 channels
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    rs = np.random.RandomState(0)
    a = np.maximum(rs.standard_normal((rows, K)), 0).astype(np.float32) * rs.uniform(0.05, 3.0, (rows, 1)).astype(np.float32)  # ReLU maps
    w = (rs.standard_normal((rows, K)) / np.sqrt(K)).astype(np.float32)
    exact = (a.astype(np.float64) * w.astype(np.float64)).sum(1)
    norm = (np.abs(a.astype(np.float64)) * np.abs(w.astype(np.float64))).sum(1)
    res = {}
    f64 = lambda t: t.astype(np.float64)
    ab, wb = to_bf16(a), to_bf16(w)
    res["bf16       (1.00)"] = (f64(ab) * f64(wb)).sum(1)
    ah, wh = a.astype(np.float16), w.astype(np.float16)
    res["fp16       (1.00)"] = (f64(ah) * f64(wh)).sum(1)
    al, wl = to_bf16(a - ab), to_bf16(w - wb)
    res["bf16x3     (3.00)"] = (f64(ab) * f64(wb) + f64(ab) * f64(wl) + f64(al) * f64(wb)).sum(1)
    la, lw = a - ah.astype(np.float32), w - wh.astype(np.float32)
    for fmt, cost in (("mx8", 2.0), ("mx6", 1.53), ("mx4", 1.53)):
        cross = (mx(la, fmt) * mx(w, fmt) + mx(a, fmt) * mx(lw, fmt)).sum(1)
        res[f"fp16+{fmt}x2 ({cost:.2f})"] = (f64(ah) * f64(wh)).sum(1) + cross
    print(f"K = {K}, {rows} dot products; error relative to sum |a||w| (what the accumulated rounding scales with)")
    print(f"{'scheme (MFMA passes / product)':34s} {'rms':>10s} {'max':>10s}   rms vs bf16x3")
    base = None
    for name, v in res.items():
        e = (v - exact) / norm
        rms, mxe = float(np.sqrt((e ** 2).mean())), float(np.abs(e).max())
        if name.startswith("bf16x3"):
            base = rms
    for name, v in res.items():
        e = (v - exact) / norm
        rms, mxe = float(np.sqrt((e ** 2).mean())), float(np.abs(e).max())
        print(f"{name:34s} {rms:10.2e} {mxe:10.2e}   {rms / base:8.1f}x")


if __name__ == "__main__":
    main()
