#!/opt/conda/bin/python3.9
"""crop84.npz against the reference's RandomCrop under the image's GENUINE scikit-image (0.18.3 in /opt/conda; the
interpreter the rest of this repository runs on has none, which is why make_goldens.py stands
``numpy.lib.stride_tricks.sliding_window_view`` in for ``skimage.util.shape.view_as_windows``).

Build-container check only (it reads /root/reference, which never travels to the GPU box):
    /opt/conda/bin/python3.9 tests/golden/check_crop_with_skimage.py
The conda interpreter has no torch and no kornia; ``augmentations.py`` imports both at module level but RandomCrop uses
neither (augmentations.py:19-75: NumPy + view_as_windows only), so the two names are stubbed -- no arithmetic of the crop
goes through a stub.  Exit status 0 = the fixture's index stream, crop bytes (sha256 + the first two crops) and centre
crop are what the reference produces with the real library."""
import hashlib
import os
import sys
import types

import numpy as np

REF = os.environ.get("CURLA_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    import skimage
    from skimage.util.shape import view_as_windows  # noqa: F401  (the genuine one: what the reference imports)
    for name in ("torch", "kornia"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    ka = types.ModuleType("kornia.augmentation")
    sys.modules["kornia"].augmentation = ka
    sys.modules["kornia.augmentation"] = ka
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import augmentations
    assert augmentations.view_as_windows is view_as_windows, "the reference did not pick up the genuine skimage"
    g = np.load(os.path.join(HERE, "crop84.npz"))
    aug = augmentations.RandomCrop((84, 84))
    assert tuple(aug.output_shape) == tuple(g["default_shape_84"])
    assert tuple(augmentations.RandomCrop((90, 160)).output_shape) == tuple(g["default_shape_90_160"])
    aug.output_shape = (76, 76)  # BASELINE.json's crop (the shipped factor 0.84 gives 71: augmentations.py:23-24)
    imgs = np.random.RandomState(int(g["imgs_seed"])).randint(0, 256, (16, 9, 84, 84), dtype=np.uint8)
    np.random.seed(int(g["numpy_seed"]))
    calls = []
    orig = np.random.randint

    def randint(*a, **k):
        r = orig(*a, **k)
        calls.append(np.array(r).copy())
        return r
    np.random.randint = randint
    try:
        out = np.ascontiguousarray(aug.training_augmentation(imgs))
    finally:
        np.random.randint = orig
    ok = True
    checks = [("h1", np.array_equal(calls[0], g["h1"])), ("w1", np.array_equal(calls[1], g["w1"])),
              ("sha256 of the 16 crops", hashlib.sha256(out.tobytes()).hexdigest() == str(g["out_sha256"])),
              ("first two crops", np.array_equal(out[:2], g["out_first2"])),
              ("centre crop", np.array_equal(np.ascontiguousarray(aug.evaluation_augmentation(imgs[0])), g["center_crop0"]))]
    for what, good in checks:
        print(("ok   " if good else "FAIL ") + what)
        ok = ok and good
    print(f"scikit-image {skimage.__version__}, numpy {np.__version__}: crop84.npz " + ("confirmed" if ok else "DIFFERS"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
