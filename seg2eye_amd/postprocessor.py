"""Image post-processing of the validation / inference path (reference data/postprocessor.py:9-130), on the device.

Only what the Tester and the OpenEDS metric use: [-1, 1] -> 0..255 with the reference's int TRUNCATION (`unnormalize`
:58-73), and the cv2.INTER_LINEAR resize to 400 x 640 followed by it (`to_255resized_imagebatch` :92-107), both through
`s2e_resize_to255` / plain tensor ops on the GPU instead of a numpy + cv2 round trip per image."""
import torch

from . import ops


class ImageProcessor:
    eps = 1e-6

    @classmethod
    def as_batch(cls, image, as_tensor=True):
        """postprocessor.py:23-45 for tensors: (H,W) / (C,H,W) -> (1,C,H,W)."""
        image = torch.as_tensor(image)
        while image.dim() < 4:
            image = image.unsqueeze(0)
        return image

    @classmethod
    def unnormalize(cls, image, as_tensor=True):
        """postprocessor.py:58-73: [-1,1] -> (x+1)*255/2; a label map (0..3) -> x/3*255; 0..255 stays; then `.int()`."""
        image = torch.as_tensor(image)
        lo, hi = float(image.min()), float(image.max())
        if lo >= -1 - cls.eps and hi <= 1 + cls.eps:
            image = torch.div(torch.mul(torch.add(image.float(), 1), 255), 2)
        elif lo >= 0 and hi < 4:
            image = torch.div(image.float(), 3) * 255
        elif lo >= 0 and hi <= 255:
            pass
        else:
            raise ValueError('Invalid ranges for image. Min: %s, max: %s' % (lo, hi))
        return image.int()

    @classmethod
    def to_255imagebatch(cls, image, as_tensor=True):
        return cls.unnormalize(cls.as_batch(image))

    @classmethod
    def to_255resized_imagebatch(cls, image, w=400, h=640, as_tensor=True):
        """(N,1,H,W) in [-1,1] on the GPU -> uint8 (N,1,h,w): bilinear resize, then the int truncation (one kernel)."""
        image = cls.as_batch(image)
        cls.assert_range1(image)
        return ops.resize_to255(image, w, h)

    @classmethod
    def assert_range1(cls, img):
        lo, hi = float(img.min()), float(img.max())
        assert lo >= -1 - cls.eps, 'Invalid ranges for image. Min: %s, max: %s' % (lo, hi)
        assert hi <= 1 + cls.eps, 'Invalid ranges for image. Min: %s, max: %s' % (lo, hi)
