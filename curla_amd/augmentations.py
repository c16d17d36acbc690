"""Observation augmentations on the learner path (reference: augmentations.py).

``RandomCrop`` keeps the reference's host-side NumPy index stream (the crop
offsets are drawn with ``np.random.randint`` exactly as augmentations.py:66-67
does, so a seeded run picks the same windows); the pixel movement itself is
fused into the first conv kernel's load (curla_amd/csrc/conv.hip) or, for
callers that want tensors, done by ``curla_crop_nchw``.
"""
import numpy as np


class IdentityAugmentation:
    """augmentations.py:7-17."""

    def __init__(self, input_shape):
        assert len(input_shape) == 2, "Input shape must be 2D"
        self.input_shape = tuple(input_shape)
        self.output_shape = tuple(input_shape)

    def evaluation_augmentation(self, image):
        return image

    def training_augmentation(self, image_batch):
        return image_batch


class RandomCrop(IdentityAugmentation):
    """augmentations.py:20-75.  ``output_shape`` may be given explicitly (the
    reference hard-codes ceil(0.84 * side), which maps 84 -> 71; BASELINE.json's
    84 -> 76 needs the override)."""

    def __init__(self, input_shape, output_shape=None):
        super().__init__(input_shape)
        self.cropping_factor = 0.84
        if output_shape is None:
            output_shape = tuple(int(np.ceil(x * self.cropping_factor)) for x in self.input_shape)
        self.output_shape = tuple(output_shape)

    def evaluation_augmentation(self, image):
        """Center crop of a (C, H, W) image (augmentations.py:26-45)."""
        h, w = self.input_shape
        new_h, new_w = self.output_shape
        top = (h - new_h) // 2
        left = (w - new_w) // 2
        return image[:, top:top + new_h, left:left + new_w]

    def draw_offsets(self, n):
        """The two RNG draws of training_augmentation (augmentations.py:66-67):
        h1 then w1, upper bounds exclusive."""
        crop_max_h = self.input_shape[0] - self.output_shape[0]
        crop_max_w = self.input_shape[1] - self.output_shape[1]
        h1 = np.random.randint(0, crop_max_h, n)
        w1 = np.random.randint(0, crop_max_w, n)
        return h1, w1

    def training_augmentation(self, image_batch):
        """Host-side crop of a (B, C, H, W) array, for callers outside the fused
        path (same result as augmentations.py:47-75: out[b] = in[b, :, h1:h1+h, w1:w1+w])."""
        n = image_batch.shape[0]
        h1, w1 = self.draw_offsets(n)
        oh, ow = self.output_shape
        out = np.empty(image_batch.shape[:2] + (oh, ow), dtype=image_batch.dtype)
        for b in range(n):
            out[b] = image_batch[b, :, h1[b]:h1[b] + oh, w1[b]:w1[b] + ow]
        return out


def make_augmentor(name, input_shape, output_shape=None):
    """augmentations.py:208-221.  ``color_jiggle`` / ``noisy_cover`` need kornia
    (not vendored by the reference, parity unpinned -- SURVEY.md D9) and are not
    part of this build yet."""
    print(f'CHOSEN AUGMENTATION: {name}')
    if name == 'identity':
        return IdentityAugmentation(input_shape)
    if name == 'random_crop':
        return RandomCrop(input_shape, output_shape)
    if name in ('color_jiggle', 'noisy_cover'):
        raise NotImplementedError(f'augmentation {name} is not available in curla_amd yet')
    raise ValueError('augmentation is not supported: %s' % name)
