"""ORACLE (test infrastructure, not product code): torch-CPU restatement of the
reference's homography estimators and the helpers around them.

Citations: /root/reference/pytracking/utils/least_squares_H.py (LSQ) and
/root/reference/pytracking/utils/geom_utils.py.

Third-party arithmetic: LSQ calls kornia==0.5.11 (README.org:28)
`kornia.geometry.epipolar.normalize_points` and
`kornia.geometry.conversions.convert_points_{to,from}_homogeneous`; kornia is not in
/root/reference nor in the image.  Their published semantics are restated below and
**parity for those three functions is unpinned** (no reference test or fixture holds
their outputs); everything else in this file is pinned against the imported reference
through tests/golden/hfit_*.npz.
"""
import numpy as np
import torch


# ---- kornia 0.5.x restatement (parity unpinned) -----------------------------------
def to_homogeneous(p):
    return torch.nn.functional.pad(p, [0, 1], "constant", 1.0)


def from_homogeneous(p, eps=1e-8):
    z = p[..., -1:]
    scale = torch.where(z.abs() > eps, 1.0 / (z + eps), torch.ones_like(z))
    return scale * p[..., :-1]


def normalize_points(points, eps=1e-8):
    """Hartley normalisation: mean-centre, mean distance sqrt(2).  (B,N,2) -> (B,N,2),(B,3,3)."""
    mean = points.mean(dim=1, keepdim=True)
    scale = (points - mean).norm(dim=-1).mean(dim=-1)
    scale = torch.sqrt(torch.tensor(2.0)) / (scale + eps)
    one, zero = torch.ones_like(scale), torch.zeros_like(scale)
    T = torch.stack([scale, zero, -scale * mean[..., 0, 0],
                     zero, scale, -scale * mean[..., 0, 1],
                     zero, zero, one], dim=-1).view(-1, 3, 3)
    pn = from_homogeneous(torch.matmul(T.unsqueeze(1), to_homogeneous(points).unsqueeze(-1)).squeeze(-1))
    return pn, T


# ---- the system  (LSQ:165-195, 299-321) ---------------------------------------------
def build_system(points1, points2, weights):
    p1, T1 = normalize_points(points1)
    p2, T2 = normalize_points(points2)
    x1, y1 = p1[..., 0:1], p1[..., 1:2]
    x2, y2 = p2[..., 0:1], p2[..., 1:2]
    one, zero = torch.ones_like(x1), torch.zeros_like(x1)
    ax = torch.cat([zero, zero, zero, -x1, -y1, -one, y2 * x1, y2 * y1], dim=-1)
    ay = torch.cat([x1, y1, one, zero, zero, zero, -x2 * x1, -x2 * y1], dim=-1)
    B, N = x1.shape[:2]
    A = torch.stack([ax, ay], dim=2).reshape(B, 2 * N, 8)          # rows interleaved (LSQ:178)
    b = torch.stack([-y2, x2], dim=2).reshape(B, 2 * N, 1)
    if weights is not None:
        w = weights[:, :, None].repeat(1, 1, 2).reshape(B, 2 * N, 1)   # plain w, not sqrt(w) (LSQ:187-193)
        A, b = w * A, w * b
    return A, b, T1, T2


def _qr_solve(A, b):
    Q, R = torch.linalg.qr(A)
    return torch.linalg.solve_triangular(R, Q.transpose(-1, -2) @ b, upper=True)


def _finish(sol, T1, T2, eps=1e-8):
    sol = torch.cat([sol, torch.ones((sol.shape[0], 1, 1), dtype=sol.dtype)], dim=1)
    H = sol.view(-1, 3, 3)
    H = T2.inverse() @ (H @ T1)
    return H / (H[..., -1:, -1:] + eps)


def _check(points1, points2):
    if points1.shape != points2.shape:
        raise AssertionError(points1.shape)
    if not (len(points1.shape) >= 1 and points1.shape[-1] == 2):
        raise AssertionError(points1.shape)
    if points1.shape[1] < 4:
        raise AssertionError(points1.shape)


def find_homography_nonhomogeneous_QR(points1, points2, weights=None):
    """LSQ:142-210."""
    _check(points1, points2)
    A, b, T1, T2 = build_system(points1, points2, weights)
    return _finish(_qr_solve(A, b), T1, T2)


def IRLSq_L1(residuals, eps=1e-8):
    """LSQ:268-269."""
    return 1 / (torch.abs(residuals) + eps)


def IRLSq_Huber(residuals, k=1, eps=1e-8):
    """LSQ:272-277."""
    a = torch.abs(residuals)
    w = 1 / (a + eps)
    w[a < k] = 1
    return w


def find_homography_IRLSq_QR(points1, points2, weights=None, reweighting_fn=IRLSq_L1, n_iter=5):
    """LSQ:280-346 (without the is_cuda assertion, LSQ:292-293)."""
    _check(points1, points2)
    A, b, T1, T2 = build_system(points1, points2, weights)
    rew = torch.ones_like(b)
    for _ in range(n_iter + 1):
        sol = _qr_solve(rew * A, rew * b)
        rew = torch.sqrt(reweighting_fn(A @ sol - b))
    return _finish(sol, T1, T2)


def torch_proj_errors(H, pts_A, pts_B):
    """LSQ:474-489.  H (B,3,3); pts (B,2,N) -> (B,N) L2 distances of H*A to B."""
    proj = torch.matmul(H, to_homogeneous(pts_A.permute(0, 2, 1)).permute(0, 2, 1))
    proj = from_homogeneous(proj.permute(0, 2, 1)).permute(0, 2, 1)
    return torch.sqrt(torch.square(proj - pts_B).sum(dim=1))


def redet_success(H, template_coords, cur_coords, thr=5.0, frac=0.2):
    """configs/YAOFT_single_control_repRAFT_sub500_noreliableinl_wLSq.py:14-21."""
    errs = torch_proj_errors(H, cur_coords[None], template_coords[None])
    return bool((errs <= thr).float().mean() > frac)


# ---- Sobol-500 subsampler (configs/..._wLSq.py:31-53) ---------------------------------
def sobol_subsample_mask(n_pts, to_draw=500):
    """Boolean mask of the picked indices (duplicates collapse, original order kept)."""
    if to_draw >= n_pts:
        return np.ones(n_pts, dtype=bool)
    mask = np.zeros(n_pts) > 0
    eng = torch.quasirandom.SobolEngine(dimension=1)
    idx = np.round(n_pts * eng.draw(to_draw).cpu().numpy().flatten()).astype(np.int32)
    mask[idx] = True
    return mask


def compose_H(*Hs):
    """geom_utils.py:365-373: product in reverse order, normalised to h33 = 1."""
    out = np.eye(3)
    for Hm in Hs:
        out = np.dot(Hm, out)
    return out / out[2, 2]
