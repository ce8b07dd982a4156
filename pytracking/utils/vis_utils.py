"""The two overlay helpers WOFT_demo.py uses (reference utils/vis_utils.py:593-621, 316-369),
written on numpy so the demo loop does not need OpenCV for them."""
import numpy as np


def blend_mask(img, mask, color=(0, 255, 0), alpha=0.5, fill=True, contour_thickness=2):
    """Overlay a binary mask: filled blend, or only its outline when fill=False."""
    out = img.copy()
    m = np.asarray(mask) > 0
    if fill:
        out[m] = (alpha * np.asarray(color) + (1 - alpha) * out[m]).astype(out.dtype)
        return out
    edge = np.zeros_like(m)
    edge[1:, :] |= m[1:, :] != m[:-1, :]
    edge[:, 1:] |= m[:, 1:] != m[:, :-1]
    t = max(int(contour_thickness) // 2, 0)
    if t:
        e = edge.copy()
        for dy in range(-t, t + 1):
            for dx in range(-t, t + 1):
                e |= np.roll(np.roll(edge, dy, 0), dx, 1)
        edge = e
    out[edge] = color
    return out


def draw_text(img, text, pos="tl", size=1, thickness=2, color=(255, 255, 255)):
    try:
        import cv2
        org = (10, 30) if pos == "tl" else (10, img.shape[0] - 10)
        return cv2.putText(img.copy(), str(text), org, cv2.FONT_HERSHEY_SIMPLEX, size, color, thickness)
    except ImportError:
        return img
