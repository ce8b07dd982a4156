"""GPU tests of the H-fit OPERATOR API (`pytracking.utils.least_squares_H` = woft_amd.homography), called the way
reference-format config files call it (configs/..._wLSq.py:24-28, ..._wIRLSq.py:24-31): argument checks and error
behaviour of least_squares_H.py:158-163,286-293, batches, the built-in losses written exactly as the reference's
configs write them, ARBITRARY re-weighting callables (least_squares_H.py:280,337), the streaming multi-workgroup
fit used for N > 8192, torch_proj_errors and compose_H -- against the reference's golden vectors and the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import hfit_ref  # noqa: E402  (checker only)
import pytracking.utils.least_squares_H as L  # noqa: E402  (the shim's import path, as configs use it)
from pytracking.utils.geom_utils import compose_H  # noqa: E402


def _corner_err(Ha, Hb):
    c = np.array([[100, 80, 1], [1800, 80, 1], [1800, 1000, 1], [100, 1000, 1.0]]).T
    pa, pb = np.asarray(Ha, np.float64) @ c, np.asarray(Hb, np.float64) @ c
    return np.abs(pa[:2] / pa[2] - pb[:2] / pb[2]).max()


def _case(golden_dir, case):
    g = np.load(golden_dir / "hfit.npz")
    return g, tuple(torch.from_numpy(g[f"{case}_{k}"]).cuda() for k in "abw")


def _synthetic(n, seed=0, outliers=0.1):
    rs = np.random.RandomState(seed)
    Hgt = np.array([[1.02, 0.03, 12.0], [-0.02, 0.98, -7.0], [2e-5, -1e-5, 1.0]])
    a = np.stack([rs.uniform(100, 1800, n), rs.uniform(80, 1000, n)], 1)
    ah = np.concatenate([a, np.ones((n, 1))], 1) @ Hgt.T
    b = ah[:, :2] / ah[:, 2:] + rs.normal(0, 0.3, (n, 2))
    no = int(outliers * n)
    b[:no] += rs.uniform(-80, 80, (no, 2))
    w = rs.uniform(0.05, 1.0, n)
    w[:no] *= 0.2
    f = lambda x: torch.from_numpy(x.astype(np.float32))[None]
    return f(a), f(b), f(w)


def test_operator_shapes_and_assertions(golden_dir):
    g, (a, b, w) = _case(golden_dir, "n500")
    H = L.find_homography_nonhomogeneous_QR(a, b, w)
    assert tuple(H.shape) == (1, 3, 3) and H.is_cuda and H.dtype == torch.float32
    assert abs(float(H[0, 2, 2]) - 1.0) < 1e-6
    assert _corner_err(H[0].cpu().numpy(), g["n500_qr_w"][0]) < 0.05
    assert _corner_err(L.find_homography_nonhomogeneous_QR(a, b)[0].cpu().numpy(), g["n500_qr_now"][0]) < 0.05
    with pytest.raises(AssertionError):                      # fewer than 4 correspondences, least_squares_H.py:162
        L.find_homography_nonhomogeneous_QR(a[:, :3], b[:, :3], w[:, :3])
    with pytest.raises(AssertionError):                      # shape mismatch, :158-159
        L.find_homography_nonhomogeneous_QR(a, b[:, :-1], w)
    with pytest.raises(AssertionError):                      # last dimension must be 2, :160-161
        L.find_homography_nonhomogeneous_QR(torch.zeros(1, 8, 3).cuda(), torch.zeros(1, 8, 3).cuda())
    with pytest.raises(AssertionError):                      # the IRLS estimator insists on device tensors, :292-293
        L.find_homography_IRLSq_QR(a.cpu(), b.cpu(), w.cpu())
    # ... the plain QR estimator does not (least_squares_H.py:142-210 has no device check): host tensors in, host tensor out -- fitted
    # by the same kernel (there is no CPU solver), so the result is the device call's, bit for bit
    Hc = L.find_homography_nonhomogeneous_QR(a.cpu(), b.cpu(), w.cpu())
    assert not Hc.is_cuda and tuple(Hc.shape) == (1, 3, 3) and torch.equal(Hc, H.cpu())
    assert torch.equal(L.find_homography_nonhomogeneous_QR(a.cpu(), b.cpu()), L.find_homography_nonhomogeneous_QR(a, b).cpu())
    with pytest.raises(AssertionError):
        L.find_homography_IRLSq_QR(a[:, :3], b[:, :3], w[:, :3])


def test_batch_of_two(golden_dir):
    g, (a, b, w) = _case(golden_dir, "n500")
    a2 = torch.cat([a, a.flip(1)], 0)
    b2 = torch.cat([b, b.flip(1)], 0)
    w2 = torch.cat([w, torch.ones_like(w)], 0)
    H = L.find_homography_nonhomogeneous_QR(a2, b2, w2)
    assert tuple(H.shape) == (2, 3, 3)
    assert _corner_err(H[0].cpu().numpy(), g["n500_qr_w"][0]) < 0.05
    assert _corner_err(H[1].cpu().numpy(), g["n500_qr_now"][0]) < 0.05      # unit weights, permuted rows
    H = L.find_homography_IRLSq_QR(a2, b2, w2, reweighting_fn=lambda r: L.IRLSq_Huber(r, k=2))
    assert _corner_err(H[0].cpu().numpy(), g["n500_irls_huber2"][0]) < 0.05


@pytest.mark.parametrize("case", ["n500", "n4096", "degen"])
def test_losses_as_the_reference_configs_write_them(golden_dir, case):
    """configs/..._wIRLSq.py:24-31: `def reweight(residuals): return IRLSq_Huber(residuals, k=2)`; the function
    default is IRLSq_L1 (least_squares_H.py:280).  Recognised through the probe -> one launch."""
    g, (a, b, w) = _case(golden_dir, case)
    tol = 0.05 if case != "degen" else 5.0                  # degenerate set: ill-conditioned by construction

    def reweight(residuals):
        return L.IRLSq_Huber(residuals, k=2)
    H = L.find_homography_IRLSq_QR(a, b, weights=w, reweighting_fn=reweight)
    assert _corner_err(H[0].cpu().numpy(), g[f"{case}_irls_huber2"][0]) < tol
    assert torch.equal(H, L.find_homography_nonhomogeneous_QR(a, b, w)) or case == "degen"   # k=2 never bites
    if case != "degen":
        H = L.find_homography_IRLSq_QR(a, b, weights=w)
        assert _corner_err(H[0].cpu().numpy(), g[f"{case}_irls_l1"][0]) < 0.2
        H = L.find_homography_IRLSq_QR(a, b, w, reweighting_fn=lambda r: L.IRLSq_Huber(r, k=0.01))
        assert _corner_err(H[0].cpu().numpy(), g[f"{case}_irls_huber001"][0]) < 0.2
    # the degenerate golden case on the weighted-LSq estimator too
    H = L.find_homography_nonhomogeneous_QR(a, b, w)
    assert _corner_err(H[0].cpu().numpy(), g[f"{case}_qr_w"][0]) < tol


def test_loss_functions_on_tensors(golden_dir):
    g = np.load(golden_dir / "hfit.npz")
    r = torch.from_numpy(g["huber_in"]).cuda()
    assert np.array_equal(L.IRLSq_Huber(r.clone(), k=1).cpu().numpy(), g["huber_k1"])
    assert np.array_equal(L.IRLSq_L1(r.clone()).cpu().numpy(), g["l1"])


def test_arbitrary_reweighting_callable(golden_dir):
    """Any callable on the residual tensor (least_squares_H.py:280,337), here spelled without the library's losses:
    (a) L1 and Huber re-written by hand must reproduce the built-in single-launch path; (b) a Cauchy loss must match
    the oracle's IRLS loop driven by the same callable; (c) the callable sees what the reference hands it."""
    g, (a, b, w) = _case(golden_dir, "n4096")
    seen = []

    def my_l1(res):
        seen.append((tuple(res.shape), res.is_cuda, res.dtype))
        return 1.0 / (res.abs() + 1e-8)
    H = L.find_homography_IRLSq_QR(a, b, w, reweighting_fn=my_l1)
    assert seen == [((1, 2 * 4096, 1), True, torch.float32)] * 6          # n_iter + 1 calls, (B, 2N, 1) residuals
    Hb = L.find_homography_IRLSq_QR(a, b, w)
    assert _corner_err(H[0].cpu().numpy(), Hb[0].cpu().numpy()) < 1e-3
    assert _corner_err(H[0].cpu().numpy(), g["n4096_irls_l1"][0]) < 0.2

    def my_huber(res, k=0.01):
        r = res.abs()
        return torch.where(r < k, torch.ones_like(r), 1.0 / (r + 1e-8))
    H = L.find_homography_IRLSq_QR(a, b, w, reweighting_fn=my_huber, n_iter=5)
    assert _corner_err(H[0].cpu().numpy(), g["n4096_irls_huber001"][0]) < 0.2

    cauchy = lambda res: 1.0 / (1.0 + (res / 0.02) ** 2)
    H = L.find_homography_IRLSq_QR(a, b, w, reweighting_fn=cauchy, n_iter=4)
    Ho = hfit_ref.find_homography_IRLSq_QR(a.cpu(), b.cpu(), w.cpu(), reweighting_fn=cauchy, n_iter=4)
    assert _corner_err(H[0].cpu().numpy(), Ho[0].numpy()) < 0.05
    # the loss matters on this data: it moves the corners away from the plain weighted fit
    assert _corner_err(H[0].cpu().numpy(), g["n4096_qr_w"][0]) > 0.05
    # first residuals = residuals of the weighted LSq solution on the normalised, weighted system (oracle restatement)
    first = []
    L.find_homography_IRLSq_QR(a, b, w, reweighting_fn=lambda r: (first.append(r.clone()), torch.ones_like(r))[1], n_iter=0)
    ro = []
    hfit_ref.find_homography_IRLSq_QR(a.cpu(), b.cpu(), w.cpu(),
                                      reweighting_fn=lambda r: (ro.append(r.clone()), torch.ones_like(r))[1], n_iter=0)
    assert float((first[0].cpu() - ro[0]).abs().max()) < 2e-4 * max(1.0, float(ro[0].abs().max()))


@pytest.mark.parametrize("n", [8193, 20000, 300001])
def test_streaming_fit_equals_single_workgroup_fit(n):
    """N > 8192 runs as the multi-workgroup pipeline (hfit_sum / dist / gram / solve): same arithmetic as the
    one-workgroup kernel (ws=None forces it), so the two agree to fp64 summation-order noise; both match the oracle."""
    from woft_amd import ops
    a, b, w = _synthetic(n, seed=n)
    pa, pb, pw = a[0].contiguous().cuda(), b[0].contiguous().cuda(), w[0].contiguous().cuda()
    out = {}
    for name, ws in (("single", False), ("stream", True)):
        for kw in (dict(), dict(reweight=1, n_irls=5), dict(reweight=2, huber_k=0.01, n_irls=5)):
            Hd, st = torch.zeros(9, device="cuda"), torch.full((1,), 7, dtype=torch.int32, device="cuda")
            import woft_amd._lib as _lib
            lib = _lib.load()
            _lib.check(lib.woft_hfit(pa.data_ptr(), pb.data_ptr(), pw.data_ptr(), n, None, kw.get("reweight", 0),
                                     float(kw.get("huber_k", 1.0)), kw.get("n_irls", 0),
                                     ops.hfit_ws().data_ptr() if ws else None, Hd.data_ptr(), st.data_ptr(),
                                     _lib.stream_ptr()), "woft_hfit")
            torch.cuda.synchronize()
            assert int(st.item()) == 0
            out[name, tuple(sorted(kw.items()))] = Hd.cpu().numpy().reshape(3, 3)
    for (name, key), H in out.items():
        if name == "stream":
            assert _corner_err(H, out["single", key]) < 1e-3, key
    Ho = hfit_ref.find_homography_nonhomogeneous_QR(a, b, w)[0].numpy()
    assert _corner_err(out["stream", ()], Ho) < 0.05
    if n <= 20000:
        Ho = hfit_ref.find_homography_IRLSq_QR(a, b, w)[0].numpy()
        assert _corner_err(out["stream", (("n_irls", 5), ("reweight", 1))], Ho) < 0.2
    # operator level: the same fit through the public function picks the streaming path by itself
    H = L.find_homography_nonhomogeneous_QR(a.cuda(), b.cuda(), w.cuda())[0].cpu().numpy()
    assert np.array_equal(H, out["stream", ()])
    # device-side count below the single-workgroup limit, and below 4 points
    cnt = torch.tensor([3], dtype=torch.int32, device="cuda")
    Hd, st = torch.zeros(9, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.hfit(pa, pb, pw, Hd, st, count=cnt)
    torch.cuda.synchronize()
    assert int(st.item()) == 1 and bool(torch.isnan(Hd).all())


def test_proj_errors_and_compose(golden_dir):
    g, (a, b, w) = _case(golden_dir, "n500")
    Hq = torch.from_numpy(g["n500_qr_w"]).cuda()
    e = L.torch_proj_errors(Hq, a.permute(0, 2, 1), b.permute(0, 2, 1))
    assert tuple(e.shape) == (1, 500)
    assert np.allclose(e.cpu().numpy(), g["n500_projerr"], rtol=1e-5, atol=1e-4)
    assert np.allclose(compose_H(g["compose_in1"], g["compose_in2"]), g["compose_12"], rtol=0, atol=1e-12)
    assert np.allclose(compose_H(g["compose_in1"], g["compose_in2"], g["compose_in1"]), g["compose_121"], rtol=0, atol=1e-12)
    assert compose_H(g["compose_in1"], None) is None
