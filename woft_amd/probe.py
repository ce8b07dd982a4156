"""Which solver back end a tracker config gets: behavioural probing of its callables.

The reference's tracker configs define their estimator, subsampler and re-detection test INLINE, as plain functions
(/root/reference/pytracking/configs/YAOFT_single_control_repRAFT_sub500_noreliableinl_wLSq.py:14-53 -- the file WOFT.py links
to, the default of WOFT_demo.py:22).  The tracker's device back end (masking -> compaction -> Sobol draw -> H fit -> inlier test
as HIP kernels, one device->host read per flow) used to be taken only by configs built from `woft_amd.presets` (tagged
callables); an unmodified reference config fell back to calling its functions on compacted device tensors (round-4 review:
"the speed of the real drop-in is unmeasured").  This module decides by what the callables DO, at tracker construction:

  * subsampler       -- called on synthetic index tensors of several sizes and weightings: accepted as "the first n points of
                        the 1-D Sobol sequence, rank = round(N u_k)" when every output is exactly that draw, in order, for every
                        probe, independently of the weights;
  * estimator        -- called under a RECORDER that stands in for the library functions the reference's configs call
                        (`pytracking.utils.least_squares_H.find_homography_nonhomogeneous_QR / find_homography_IRLSq_QR`, here
                        woft_amd.homography): accepted when it makes exactly one library call, hands its own arguments through
                        untouched and returns that call's result untouched; an IRLS re-weighting callable is matched against
                        IRLSq_L1 / IRLSq_Huber(k) on a residual sweep;
  * re-detection test -- the same recorder answers its `torch_proj_errors` call with crafted error vectors: accepted when the
                        verdict is `mean(errs <= thr) > frac` for a (thr, frac) pair identified by bisection and then checked
                        on a grid of (inlier count, N) cases, and when it projects the current-frame points onto the template.

Anything else -- a function that post-processes its inputs, calls nothing or something else, draws differently, raises on the
probe inputs -- keeps the callable back end, where the config's own functions run exactly as TRK:141-162 runs them.  Both back
ends produce the same homographies on equivalent callables (tests/test_tracker_gpu.py); the probe only picks the faster one.
`WOFT_FUSED=0` / config key `device_solver = False` turn the device back end off altogether.
"""
import threading

import numpy as np
import torch

_TLS = threading.local()


def recorder():
    """The active _Recorder of this thread, or None (woft_amd.homography consults it at the top of its entry points)."""
    return getattr(_TLS, "rec", None)


class _Recorder:
    def __init__(self, proj_answer=None):
        self.calls = []
        self.proj_answer = proj_answer

    def __enter__(self):
        self._prev = getattr(_TLS, "rec", None)
        _TLS.rec = self
        return self

    def __exit__(self, *exc):
        _TLS.rec = self._prev
        return False

    # -- stand-ins (called by woft_amd.homography) ------------------------------------------------------------
    def fit(self, kind, points1, points2, weights, reweighting_fn=None, n_iter=None):
        out = torch.full((points1.shape[0], 3, 3), float("nan"), device=points1.device)
        self.calls.append(dict(kind=kind, a=points1, b=points2, w=weights, fn=reweighting_fn, n_iter=n_iter, out=out))
        return out

    def proj(self, H, pts_a, pts_b):
        self.calls.append(dict(kind="proj", H=H, a=pts_a, b=pts_b))
        if self.proj_answer is None:
            return torch.zeros(pts_a.shape[0], pts_a.shape[2], device=pts_a.device)
        return self.proj_answer.to(pts_a.device)


# ---- subsampler ----------------------------------------------------------------------------------------------
def probe_subsampler(fn, device="cpu", max_draw=1024):
    """-> n_draw (int) when fn(coords_a, coords_b, weights, post) keeps exactly the correspondences of rank round(N * u_k),
    u = first n_draw 1-D Sobol points, in their original order (no-op for N <= n_draw), whatever the weights; else None."""
    def run(n, w):
        idx = torch.arange(n, dtype=torch.float32, device=device)
        a = torch.stack([idx, idx + 0.25])
        b = torch.stack([idx + 0.5, 2 * idx])
        out = fn(a, b, w, None)
        if not (isinstance(out, tuple) and len(out) == 4 and out[3] is None):
            return None
        oa, ob, ow = out[:3]
        if not all(isinstance(t, torch.Tensor) for t in (oa, ob, ow)) or oa.dim() != 2 or oa.shape[0] != 2:
            return None
        kept = oa[0].detach().cpu().numpy().astype(np.int64)
        k = kept.size
        ok = (tuple(ob.shape) == (2, k) and tuple(ow.shape) == (1, k)
              and np.array_equal(oa[1].cpu().numpy(), kept + 0.25) and np.array_equal(ob[0].cpu().numpy(), kept + 0.5)
              and np.array_equal(ob[1].cpu().numpy(), 2.0 * kept) and np.array_equal(ow[0].cpu().numpy(), w[0].cpu().numpy()[kept]))
        return kept if ok else None

    from .presets import sobol_points
    try:
        big = 1 << 20                      # (every rank distinct for n_draw <= 1024: the points differ by >= 2^-10)
        kept = run(big, torch.ones(1, big, device=device))
        if kept is None or kept.size > max_draw or kept.size < 1:
            return None
        n_draw = int(kept.size)
        u = sobol_points(n_draw)
        g = torch.Generator().manual_seed(11)
        for n in (big, 100003, 2 * n_draw + 1, n_draw + 1, n_draw, max(n_draw // 2, 4), 4):
            if n_draw >= n:
                want = np.arange(n)
            else:
                keep = np.zeros(n + 1, dtype=bool)
                keep[np.round(n * u).astype(np.int32)] = True
                if keep[n]:
                    return None            # (rank N: the reference's own indexing would raise there)
                want = np.flatnonzero(keep[:n])
            for w in (torch.ones(1, n), torch.zeros(1, n), torch.rand(1, n, generator=g)):
                got = run(n, w.to(device))
                if got is None or not np.array_equal(got, want):
                    return None
        return n_draw
    except Exception:
        return None


# ---- estimator -----------------------------------------------------------------------------------------------
def _match_loss(fn):
    """A re-weighting callable -> (reweight code, k): 1 = IRLSq_L1, 2 = IRLSq_Huber(k) (woft_hfit's built-in losses), by its
    values on a residual sweep (both signs, eight decades); None when it is neither."""
    from .homography import IRLSq_Huber, IRLSq_L1, _Probe
    try:                                   # a lambda around the library's own losses says so itself (homography._Probe)
        kind = fn(_Probe())
        if isinstance(kind, tuple) and len(kind) == 3 and kind[2] == 1e-8 and kind[0] in ("l1", "huber"):
            return (1, 0.0) if kind[0] == "l1" else (2, float(kind[1]))
    except Exception:
        pass
    r = torch.cat([torch.logspace(-6, 4, 400), -torch.logspace(-6, 4, 400), torch.zeros(1)]).view(1, -1, 1)
    try:
        got = fn(r.clone())
        if not isinstance(got, torch.Tensor) or got.shape != r.shape:
            return None
        if torch.equal(got, IRLSq_L1(r.clone())):
            return 1, 0.0
        # Huber: weight 1 below k, 1 / (|r| + eps) from k on: k = the smallest |r| whose weight is not 1
        a = r.abs().reshape(-1)
        off = got.reshape(-1) != 1.0
        if not bool(off.any()):
            return None
        k_hi = float(a[off].min())
        k_lo = float(a[(~off) & (a < k_hi)].max()) if bool(((~off) & (a < k_hi)).any()) else 0.0
        # candidates: "nice" thresholds inside (k_lo, k_hi] -- verified exactly below, on a sweep that straddles them
        cands = sorted({float(np.format_float_positional(x, precision=p, unique=False, fractional=False))
                        for p in (1, 2, 3) for x in (k_hi, 0.5 * (k_lo + k_hi))} | {k_hi})
        for k in cands:
            if not (k_lo < k <= k_hi):
                continue
            rr = torch.cat([r.reshape(-1), torch.tensor([k, -k, k * (1 - 1e-6), k * (1 + 1e-6)])]).view(1, -1, 1)
            if torch.equal(fn(rr.clone()), IRLSq_Huber(rr.clone(), k=k)):
                return 2, float(k)
    except Exception:
        return None
    return None


def _probe_estimator_once(fn, n, device, with_weights=True):
    g = torch.Generator().manual_seed(5 + n)
    a = torch.rand(1, n, 2, generator=g).to(device) * 100
    b = torch.rand(1, n, 2, generator=g).to(device) * 100
    w = torch.rand(1, n, generator=g).to(device) if with_weights else None
    with _Recorder() as rec:
        out = fn(a, b, w)
    if len(rec.calls) != 1:
        return None
    c = rec.calls[0]
    if c["kind"] not in ("lsq", "irls") or out is not c["out"]:
        return None
    same = lambda x, y: x is y or (isinstance(x, torch.Tensor) and isinstance(y, torch.Tensor) and x.shape == y.shape
                                   and x.dtype == y.dtype and torch.equal(x, y))
    if not (same(c["a"], a) and same(c["b"], b) and (c["w"] is None or same(c["w"], w))):
        return None
    weighted = c["w"] is not None          # (the reference's "plainLSq" configs hand the library weights=None: an unweighted fit)
    if c["kind"] == "lsq":
        return 0, 0.0, 0, weighted
    loss = _match_loss(c["fn"])
    n_iter = c["n_iter"]
    if loss is None or not isinstance(n_iter, int) or not (0 <= n_iter <= 64):
        return None
    return loss[0], loss[1], n_iter, weighted


ESTIMATOR_PROBE_SIZES = (64, 4, 5, 500, 1500)


def probe_estimator(fn, device="cpu"):
    """-> (reweight, huber_k, n_irls, weighted) when fn(pts_A (1,N,2), pts_B, weights (1,N)) is one pass-through call of the
    library's least-squares / IRLS estimator (reweight 0 / 1 L1 / 2 Huber; weighted = it hands the weights on), else None.
    Asked at several N -- the minimal 4, 5, the subsampler's 500, beyond the one-workgroup solver's range -- and every answer must be
    the same (a callable that switches estimator on the number of correspondences keeps the callable back end); where it also accepts
    weights=None it must make the same call there, unweighted."""
    try:
        first = None
        for n in ESTIMATOR_PROBE_SIZES:
            got = _probe_estimator_once(fn, n, device)
            if got is None or (first is not None and got != first):
                return None
            first = got
        try:
            plain = _probe_estimator_once(fn, 64, device, with_weights=False)
        except Exception:
            plain = "raises"                   # (it needs its weights: the tracker always provides them)
        if plain != "raises" and (plain is None or plain[:3] != first[:3] or plain[3]):
            return None
        return first
    except Exception:
        return None


# ---- re-detection test ---------------------------------------------------------------------------------------
def _f32_bits(x):
    return int(np.float32(x).view(np.int32))


def _bits_f32(i):
    return float(np.int32(i).view(np.float32))


def probe_redetection(fn, device="cpu"):
    """-> (threshold_px, min_fraction) when fn(H, template_coords (2,N), cur_coords (2,N), weights) is
    `mean(torch_proj_errors(H, cur[None], template[None]) <= threshold_px) > min_fraction`; ("const", bool) when it returns the
    same Python bool without looking at anything; else None."""
    g = torch.Generator().manual_seed(9)
    n = 1000
    H = torch.eye(3)[None].to(device)
    tmpl = (torch.rand(2, n, generator=g) * 100).to(device)
    cur = (torch.rand(2, n, generator=g) * 100).to(device)
    w = torch.rand(1, n, generator=g).to(device)

    def verdict(errs):
        with _Recorder(errs.view(1, -1)) as rec:
            v = fn(H, tmpl, cur, w)
        if len(rec.calls) != 1 or rec.calls[0]["kind"] != "proj":
            raise ValueError("not one projection-error call")
        c = rec.calls[0]
        ok = (c["H"] is H or torch.equal(c["H"], H)) and tuple(c["a"].shape) == (1, 2, n) and torch.equal(c["a"][0], cur) \
            and tuple(c["b"].shape) == (1, 2, n) and torch.equal(c["b"][0], tmpl)
        if not ok:
            raise ValueError("projection of other points")
        return bool(v)

    try:
        big = 3.0e38
        # a verdict that never looks at anything (the reference's "alwayswarp" / "neverwarp" ablations: `return True` / `return False`)
        const = set()
        trials = [(H, tmpl, cur, w), (H * 3.0, tmpl * 0.0, cur + 1e4, torch.zeros_like(w)), (H.flip(1), cur, tmpl, torch.ones_like(w)),
                  (H, tmpl[:, :7], cur[:, :7], w[:, :7] * 100.0), (-H, -tmpl, cur * 1e-3, 1.0 - w)]
        for args in trials:                # (inputs of every kind: a verdict that reads ANY of them is not a constant)
            with _Recorder(torch.zeros(1, args[1].shape[1])) as rec:
                v = fn(*args)
            if rec.calls or not isinstance(v, (bool, np.bool_)):
                const = None
                break
            const.add(bool(v))
        if const is not None:
            return ("const", const.pop()) if len(const) == 1 else None
        if not verdict(torch.zeros(n)) or verdict(torch.full((n,), big)):
            return None
        # threshold: the largest float32 t with "every error = t" still a success (positive floats order like their bit patterns)
        lo, hi = _f32_bits(0.0), _f32_bits(big)
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if verdict(torch.full((n,), _bits_f32(mid))):
                lo = mid
            else:
                hi = mid
        thr = _bits_f32(lo)
        # fraction: smallest inlier count of N0 that succeeds -> an interval for min_fraction; take the shortest decimal in it
        n0 = 1 << 20

        def frac_ok(count, total):
            e = torch.full((total,), big)
            e[:count] = 0.0
            if total != n:
                return bool(_call_sized(fn, e, total, device))
            return verdict(e)
        lo_c, hi_c = 0, n0                 # verdict(lo_c) False, verdict(hi_c) True
        if frac_ok(0, n0) or not frac_ok(n0, n0):
            return None
        while hi_c - lo_c > 1:
            mid = (lo_c + hi_c) // 2
            if frac_ok(mid, n0):
                hi_c = mid
            else:
                lo_c = mid
        f_lo, f_hi = lo_c / n0, hi_c / n0   # success iff count / N > min_fraction:  f_lo <= min_fraction < f_hi
        frac = None
        for p in range(1, 7):
            c = float(np.format_float_positional(0.5 * (f_lo + f_hi), precision=p, unique=False, fractional=False))
            if f_lo <= c < f_hi:
                frac = c
                break
        if frac is None:
            return None
        # check the rule on a grid of (count, N) around the boundary, N as the tracker produces them, and errors around thr
        for total in (4, 5, 10, 37, 100, 499, 500, 1000, 1024):
            for count in {0, total, int(frac * total) - 1, int(frac * total), int(frac * total) + 1, int(np.ceil(frac * total))}:
                if 0 <= count <= total:
                    want = np.float32(count) / np.float32(total) > np.float32(frac)
                    if frac_ok(count, total) != bool(want):
                        return None
        e = torch.full((n,), big)
        e[:n // 2] = thr
        e[n // 2:n // 2 + 10] = _bits_f32(_f32_bits(thr) + 1)
        if verdict(e) != (np.float32(n // 2) / np.float32(n) > np.float32(frac)):
            return None
        return float(thr), float(frac)
    except Exception:
        return None


def _call_sized(fn, errs, total, device):
    g = torch.Generator().manual_seed(total)
    H = torch.eye(3)[None].to(device)
    tmpl = (torch.rand(2, total, generator=g) * 100).to(device)
    cur = (torch.rand(2, total, generator=g) * 100).to(device)
    w = torch.rand(1, total, generator=g).to(device)
    with _Recorder(errs.view(1, -1)) as rec:
        v = fn(H, tmpl, cur, w)
    if len(rec.calls) != 1 or rec.calls[0]["kind"] != "proj" or not torch.equal(rec.calls[0]["a"][0], cur) \
            or not torch.equal(rec.calls[0]["b"][0], tmpl):
        raise ValueError("not the projection-error rule")
    return bool(v)


def solver_spec(estimator, subsampler, redetection, device="cpu"):
    """The three callables of a tracker config -> the device back end's parameters
    dict(reweight, huber_k, n_irls, thr, min_frac, n_draw) or None (callable back end), and how it was decided."""
    how = []
    spec = getattr(estimator, "woft_spec", None)
    est = (int(spec[1]), float(spec[2]), int(spec[3]), True) if spec is not None else probe_estimator(estimator, device)
    how.append("estimator: " + ("tagged" if spec is not None else ("probed" if est is not None else "callable")))
    spec = getattr(redetection, "woft_spec", None)
    if spec is not None:
        red = (float(spec[1]), float(spec[2])) if spec[0] == "inliers" else None
    else:
        red = probe_redetection(redetection, device)
    how.append("re-detection: " + ("tagged" if spec is not None else ("probed" if red is not None else "callable")))
    if not subsampler:
        n_draw = 0
        how.append("subsampler: none")
    else:
        spec = getattr(subsampler, "woft_spec", None)
        if spec is not None:
            n_draw = int(spec[1]) if spec[0] == "sobol" else None
        else:
            n_draw = probe_subsampler(subsampler, device)
        how.append("subsampler: " + ("tagged" if spec is not None else ("probed" if n_draw is not None else "callable")))
    if est is None or red is None or n_draw is None or n_draw > 1024:
        return None, "; ".join(how)
    const = red[1] if red[0] == "const" else None
    return dict(reweight=est[0], huber_k=est[1], n_irls=est[2], weighted=est[3], thr=5.0 if const is not None else red[0],
                min_frac=0.0 if const is not None else red[1], const_verdict=const, n_draw=n_draw), "; ".join(how)
