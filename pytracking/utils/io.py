"""Frame source with the reference's GeneralVideoCapture surface (utils/io.py:145-177): a directory
of images (sorted file names) read with OpenCV when it is installed, with PIL otherwise; BGR uint8."""
import os
from pathlib import Path

import numpy as np


_IMAGE_SUFFIXES = {".jpg", ".jpeg", ".png"}


def _imread_bgr(path):
    try:
        import cv2
        return cv2.imread(str(path))
    except ImportError:
        from PIL import Image
        return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])


class GeneralVideoCapture:
    def __init__(self, path, reverse=False):
        self.image_inputs = Path(path).is_dir()
        if self.image_inputs:
            self.path = path
            self.images = sorted(q.name for q in Path(path).iterdir() if q.is_file() and q.suffix.lower() in _IMAGE_SUFFIXES)
            if reverse:
                self.images = self.images[::-1]
            self.i = 0
        else:
            import cv2                                     # video files need OpenCV
            self.cap = cv2.VideoCapture(str(path))

    def read(self):
        if not self.image_inputs:
            return self.cap.read()
        if self.i >= len(self.images):
            return False, None
        self.frame_src = self.images[self.i]
        img = _imread_bgr(os.path.join(self.path, self.images[self.i]))
        self.i += 1
        return True, img

    def release(self):
        return None if self.image_inputs else self.cap.release()
