"""Observation augmentations on the learner path (reference: augmentations.py).

``RandomCrop`` keeps the reference's host-side NumPy index stream (the crop
offsets are drawn with ``np.random.randint`` exactly as augmentations.py:66-67
does, so a seeded run picks the same windows); the pixel movement itself is
fused into the first conv kernel's load (curla_amd/csrc/conv.hip) or, for
callers that want tensors, done by ``curla_crop_nchw``.
"""
import numpy as np


def _device_batch(image_batch, input_shape):
    """The float NCHW device tensor the augmentation kernels take; anything else is an error (no CPU path)."""
    import torch
    if not (torch.is_tensor(image_batch) and image_batch.dim() == 4 and image_batch.shape[1] % 3 == 0
            and tuple(image_batch.shape[2:]) == tuple(input_shape)):
        raise ValueError("expected a (B, 3*frame_stack, %d, %d) tensor, got %r"
                         % (input_shape[0], input_shape[1], getattr(image_batch, "shape", type(image_batch))))
    from . import _lib
    if not image_batch.is_cuda and _lib._trace_hook is None:
        raise RuntimeError("ColorJiggle / NoisyCover run on the HIP device only: curla_amd has no CPU path")
    return image_batch.float().contiguous()


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


class ColorJiggle(IdentityAugmentation):
    """augmentations.py:78-136: every RGB frame of the stack is jittered independently with probability
    0.85 -- contrast U(0.8,1.2), saturation U(0.5,1.5), hue U(-0.5,0.5) turns, brightness 0 -- the four
    operations applied in one random order per call.

    PARITY UNPINNED: the reference delegates the arithmetic to kornia (not vendored, version un-pinned).
    The arithmetic here (curla_amd/csrc/augment.hip, restated in oracle/curla_oracle.py) follows kornia's
    documented ColorJiggle: brightness additive (0 -> identity), contrast x*c clamped to [0,1], saturation
    and hue through HSV.  Random parameters are drawn on the host from torch's CPU generator."""

    p, contrast, saturation, hue = 0.85, 0.2, 0.5, 0.5

    def draw_params(self, n_images):
        """(params [n_images, 4] = apply, contrast, saturation, hue in radians; order [4])."""
        import torch
        apply = (torch.rand(n_images) < self.p).float()
        con = torch.empty(n_images).uniform_(1 - self.contrast, 1 + self.contrast)
        sat = torch.empty(n_images).uniform_(1 - self.saturation, 1 + self.saturation)
        hue = torch.empty(n_images).uniform_(-self.hue, self.hue) * (2 * np.pi)
        order = torch.randperm(4).int()
        return torch.stack([apply, con, sat, hue], 1).contiguous(), order

    def training_augmentation(self, image_batch, params=None, order=None):
        """augmentations.py:105-136 on the reference's tensor contract: a float (B, 3k, H, W) device tensor in
        [0,255] in, the jittered batch out -- the same kernel arithmetic ``ReplayBuffer`` applies straight from
        the ring.  A new tensor is returned (the reference scales its argument in place by 1/255 and returns a
        fresh tensor; callers only use the return value, utils.py:174-182).  ``params`` / ``order`` replace
        the random draws (tests)."""
        import torch
        from . import ops
        x = _device_batch(image_batch, self.input_shape)
        B, C = x.shape[:2]
        if params is None:
            params, order = self.draw_params(B * (C // 3))
        out = torch.empty_like(x)
        ops.color_jiggle_nchw(x, params.to(x.device, torch.float32).contiguous(),
                              torch.as_tensor(order, dtype=torch.int32).to(x.device), out)
        return out


class NoisyCover(IdentityAugmentation):
    """augmentations.py:138-205: rows [0, ceil(0.31 h)) and [h - ceil(0.20 h), h) of every frame are painted
    with one random colour per RGB channel (np.random.randint(0, 255) x3 per call, shared by the batch),
    Gaussian noise N(0, 10) is added and the result clamped to [0, 255].  Noise: torch generator of the
    device (kornia's RandomGaussianNoise in the reference; PARITY UNPINNED for the noise stream only)."""

    def __init__(self, input_shape):
        super().__init__(input_shape)
        self.h = self.input_shape[0]
        self.top = int(np.ceil(self.h * 0.31))
        self.bottom = int(np.ceil(self.h * 0.20))
        self.std = 10.0

    def draw_colors(self):
        return [np.random.randint(0, 255) for _ in range(3)]

    def training_augmentation(self, image_batch, colors=None, noise=None):
        """augmentations.py:170-205 on the reference's tensor contract (float (B, 3k, H, W) device tensor in
        [0,255]); returns a new tensor (the reference paints the cover into its argument in place and returns a
        fresh noisy tensor).  ``colors`` / ``noise`` replace the random draws (tests)."""
        import torch
        from . import ops
        x = _device_batch(image_batch, self.input_shape)
        if colors is None:
            colors = self.draw_colors()
        if noise is None:
            noise = torch.randn(x.shape, device=x.device) * self.std
        out = torch.empty_like(x)
        ops.noisy_cover_nchw(x, noise.to(x.device, torch.float32).contiguous(), colors, self.top, self.bottom, out)
        return out


def make_augmentor(name, input_shape, output_shape=None):
    """augmentations.py:208-221."""
    print(f'CHOSEN AUGMENTATION: {name}')
    if name == 'identity':
        return IdentityAugmentation(input_shape)
    if name == 'random_crop':
        return RandomCrop(input_shape, output_shape)
    if name == 'color_jiggle':
        return ColorJiggle(input_shape)
    if name == 'noisy_cover':
        return NoisyCover(input_shape)
    raise ValueError('augmentation is not supported: %s' % name)
