"""Call the first-layer kernel test body repeatedly in one process; print every assertion that fires."""
import sys, os, traceback
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tests import test_gpu_kernels as T
from curla_amd import _lib, ops
n_bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    for impl in ("hybrid", "band", "rw"):  # (what the u8_impl fixture does)
        _lib.set_option("conv1_u8", impl)
        for case in T.CONV1_CASES:
            try:
                T.test_crop_and_conv1_u8(ops, impl, *case)
            except AssertionError as e:
                n_bad += 1
                print("rep", rep, impl, case, "->", str(e)[:200], flush=True)
_lib.set_option("conv1_u8", "hybrid")
print("failures:", n_bad)
