"""ctypes binding of libcurla_hip.so (include/curla_hip.h).

There is no fallback: if the library is missing or a kernel reports an error
the call raises.  ``torch`` is used only to obtain device pointers and the
current HIP stream handle.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CURLA_LIB_PATH: another build of the library (an A/B partner of a measurement, a C-ABI consumer's own build); the
# in-tree library's source-stamp check does not apply to it
LIB_PATH = os.environ.get("CURLA_LIB_PATH") or os.path.join(_HERE, "libcurla_hip.so")

c_int, c_ll, c_float, c_size_t, vp = ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p
c_u64 = ctypes.c_ulonglong
c_double = ctypes.c_double

ABI_VERSION = 6  # include/curla_hip.h CURLA_ABI_VERSION this table was written for

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/curla_hip.h
SIGNATURES = {
    "curla_conv1_fwd": [vp, c_int, vp, vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, vp],
    "curla_conv3x3_s1_fwd": [vp, vp, vp, vp, c_int, c_int, c_int, c_int, vp],
    "curla_conv3x3_s1_fwd2": [vp, vp, vp, vp, c_int, vp, vp, vp, vp, c_int, c_int, c_int, c_int, vp],
    "curla_conv3x3_s1_fwd_stack": [c_int, vp, vp, vp, vp, c_int, vp, vp, vp, vp, c_int, c_int, c_int, c_int, vp],
    "curla_conv3x3_s1_stack_granule": [],
    "curla_conv1_fwd2": [vp, vp, vp, vp, vp, vp, vp, c_int, vp, vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_int,
                         c_int, c_float, vp],
    "curla_conv3x3_s1_dgrad": [vp, vp, vp, vp, c_int, c_int, c_int, c_int, vp],
    "curla_conv3x3_s1_wgrad": [vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, vp],
    "curla_conv1_wgrad": [vp, c_int, vp, vp, vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                          c_float, vp],
    "curla_conv3x3_s1_wgrad_slabs": [vp, vp, vp, c_int, c_int, c_int, c_int, vp, vp],
    "curla_conv1_wgrad_slabs": [vp, c_int, vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, vp,
                                vp],
    "curla_conv3x3_s1_bwd_slabs": [vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, vp, vp],
    "curla_wgrad_reduce_multi": [c_int, vp, vp, vp, vp, vp, vp, vp],
    "curla_conv_wgrad_workspace_floats": [c_int],
    "curla_gemm": [vp, c_int, c_int, c_ll, vp, c_int, c_int, c_ll, vp, c_int, c_ll, c_int, c_int, c_int, c_int, c_int,
                   c_ll, c_float, vp, c_ll, c_int, vp, c_int, c_ll, vp],
    "curla_fc_dx": [vp, vp, vp, vp, c_int, c_int, c_int, vp],
    "curla_fc_dw": [vp, vp, vp, c_int, c_int, c_int, vp],
    "curla_fc_bwd": [vp, vp, vp, vp, vp, c_int, c_int, c_int, vp],
    "curla_fc_bwd_ln": [vp, vp, vp, vp, vp, c_int, c_int, c_int, vp, c_int, vp, vp, vp, vp],
    "curla_fc_dw_ln": [vp, vp, vp, c_int, c_int, c_int, vp, c_int, vp, vp, vp, vp],
    "curla_fc_bwd_ln2": [vp, vp, vp, vp, vp, c_int, c_int, c_int, vp, c_int, vp, vp, vp, vp, c_int, c_int, vp, vp],
    "curla_curl_head": [vp, vp, vp, vp, vp, vp, c_int, c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "curla_gemm_nested": [vp, c_int, c_int, c_ll, c_ll, vp, c_int, c_int, c_ll, c_ll, vp, c_int, c_ll, c_ll, c_int, c_int, c_int,
                          c_int, c_int, c_float, vp, c_ll, c_ll, c_int, vp, c_int, c_ll, c_ll, vp],
    "curla_mlp_out_fwd_nested": [vp, c_ll, c_ll, vp, c_ll, c_ll, vp, c_ll, c_ll, vp, c_ll, c_ll, c_int, c_int, c_int, c_int, c_int,
                                 vp],
    "curla_gemm_multi": [c_int, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_ll, vp],
    "curla_fc_fwd_multi": [c_int, vp, vp, vp, c_int, c_int, c_int, c_int, c_ll, c_int, vp],
    "curla_splitk_reduce": [vp, c_int, c_ll, c_int, c_int, c_int, vp, c_int, vp, c_int, vp],
    "curla_mlp_out_fwd": [vp, c_ll, vp, c_ll, vp, c_ll, vp, c_ll, c_int, c_int, c_int, c_int, vp],
    "curla_mlp_out_bwd": [vp, c_ll, vp, c_ll, vp, c_ll, vp, c_ll, vp, c_ll, c_int, c_int, c_int, c_int, vp],
    "curla_mlp_out_bwd_loss": [vp, vp, c_ll, vp, c_ll, vp, c_ll, vp, c_ll, c_int, c_int, vp, vp, c_ll, vp],
    "curla_mlp_out_bwd_bias": [vp, c_ll, vp, c_ll, vp, c_ll, vp, c_ll, vp, c_ll, c_int, c_int, c_int, c_int, vp, vp, c_ll,
                               vp],
    "curla_gemm_small_shape": [c_int, c_int, c_int, c_int],
    "curla_gemm_colsum": [vp, c_int, c_int, c_ll, vp, c_int, c_int, c_ll, vp, c_int, c_ll, c_int, c_int, c_int, c_int, vp,
                          c_ll, vp],
    "curla_linear_bwd": [vp, c_ll, vp, c_ll, vp, c_ll, vp, c_ll, vp, c_ll, vp, c_ll, vp, c_ll, c_int, c_int, c_int, c_int, vp],
    "curla_fc_ln_fwd": [vp, c_int, c_ll, c_int, vp, vp, vp, c_int, c_int, c_float, vp, vp, vp, vp, c_int, vp, vp, c_int,
                        vp],
    "curla_critic_td_loss": [vp, vp, c_ll, vp, vp, vp, vp, c_float, c_int, vp, vp, vp, vp],
    "curla_soft_update2": [vp, vp, c_size_t, c_size_t, c_float, c_float, c_float, c_float, vp],
    "curla_adam_step": [vp, vp, vp, vp, c_size_t, c_double, c_double, c_double, c_double, c_ll, vp, vp],
    "curla_adam_step_lerp": [vp, vp, vp, vp, c_size_t, c_double, c_double, c_double, c_double, c_ll, vp, vp, c_size_t,
                             c_float, c_float, c_float, c_float, vp],
    "curla_adam_step_scalar64": [vp, vp, vp, vp, c_size_t, c_double, c_double, c_double, c_double, c_ll, vp, vp, vp, vp, vp,
                                 c_double, c_double, c_double, c_double, c_ll, vp, vp],
    "curla_adam_step2": [vp, vp, vp, vp, vp, vp, c_size_t, c_size_t, c_double, c_double, c_double, c_double, c_ll, c_double,
                         c_double, c_double, c_double, c_ll, vp, vp],
    "curla_f64_pack": [vp, vp, vp],
    "curla_f64_unpack": [vp, c_double, c_double, vp, vp],
    "curla_gather_transition_scalars": [vp, vp, c_int, c_int, vp, vp, vp, vp],
    "curla_sample_stage": [vp, vp, c_ll, vp, c_int, c_int, vp, vp, vp, vp],
    "curla_host_device_pointer": [vp, vp],
    "curla_ln_bwd": [vp, vp, vp, vp, c_int, c_int, vp, vp, vp, vp, vp],
    "curla_ln_bwd_twin": [vp, vp, c_int, vp, vp, vp, c_int, c_int, vp, vp, vp, vp, vp],
    "curla_ln_bwd_partial": [vp, vp, c_int, vp, vp, vp, c_int, c_int, vp, vp, vp, vp],
    "curla_colsum": [vp, c_int, c_int, c_int, c_ll, vp, c_ll, c_int, vp],
    "curla_colsum3": [vp, c_int, vp, c_int, vp, c_int, c_int, vp, vp, vp, c_ll, c_int, vp],
    "curla_actor_head_fwd": [vp, vp, c_int, c_int, c_float, c_float, vp, vp, vp, vp, vp, vp, c_int, vp],
    "curla_mlp_out_head_fwd": [vp, vp, vp, vp, c_int, c_int, c_int, vp, c_float, c_float, vp, vp, vp, vp, vp, vp, c_int, vp],
    "curla_actor_head_fwd_rng": [vp, vp, c_u64, c_u64, vp, c_int, c_int, c_float, c_float, vp, vp, vp, vp, vp, vp, c_int, vp],
    "curla_mlp_out_head_fwd_rng": [vp, vp, vp, vp, c_int, c_int, c_int, vp, c_u64, c_u64, vp, c_float, c_float, vp, vp, vp,
                                   vp, vp, vp, c_int, vp],
    "curla_fc_ln_fwd_multi": [c_int, vp, c_int, c_ll, c_int, c_int, c_int, c_float, c_int, vp],
    "curla_actor_head_bwd": [vp, vp, c_int, vp, vp, c_float, vp, vp, vp, vp, c_int, c_int, c_float, c_float, vp, vp],
    "curla_concat": [vp, vp, c_int, c_int, c_int, vp, vp],
    "curla_split_sum": [vp, c_ll, c_int, c_int, c_int, vp, vp, vp],
    "curla_td_target": [vp, c_ll, vp, vp, vp, vp, c_float, c_int, vp, vp],
    "curla_critic_loss": [vp, c_ll, vp, c_int, vp, vp, vp],
    "curla_actor_loss": [vp, c_ll, vp, vp, c_int, vp, c_float, c_int, vp, vp, vp, vp],
    "curla_curl_ce": [vp, c_int, c_int, vp, vp, vp, vp],
    "curla_mean": [vp, c_int, vp, vp],
    "curla_soft_update": [vp, vp, c_size_t, c_float, c_float, vp],
    "curla_crop_nchw": [vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, vp, vp, vp],
    "curla_store_frame": [vp, vp, c_ll, c_int, c_int, c_int, vp],
    "curla_gather_stacks": [vp, vp, c_int, vp, c_int, c_int, c_int, c_int, vp, vp],
    "curla_nhwc_to_nchw": [vp, vp, c_int, c_int, c_int, c_int, vp],
    "curla_color_jiggle": [vp, vp, vp, vp, c_int, c_int, c_int, c_int, vp, vp],
    "curla_noisy_cover": [vp, vp, vp, c_float, c_float, c_float, c_int, c_int, c_int, c_int, c_int, c_int, vp, vp],
    "curla_gather_nhwc": [vp, vp, c_int, c_int, c_int, c_int, vp, vp],
    "curla_color_jiggle_nchw": [vp, vp, vp, c_int, c_int, c_int, c_int, vp, vp],
    "curla_noisy_cover_nchw": [vp, vp, c_float, c_float, c_float, c_int, c_int, c_int, c_int, c_int, c_int, vp, vp],
    "curla_version": [],
    "curla_abi_version": [],
    "curla_set_option": [ctypes.c_char_p, ctypes.c_char_p],
    "curla_get_option": [ctypes.c_char_p],
}
_RESTYPES = {"curla_conv_wgrad_workspace_floats": c_size_t, "curla_version": ctypes.c_char_p,
             "curla_get_option": ctypes.c_char_p}
_ERRORS = {-1: "CURLA_ERR_ARG (bad pointer/size/alignment)", -2: "CURLA_ERR_LAUNCH (HIP launch failed)",
           -3: "CURLA_ERR_UNSUPPORTED (shape not supported by the gfx950 kernels)"}


class CurlaHipError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen the library; raises if it has not been built (python -m curla_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CurlaHipError(f"{LIB_PATH} is missing: build it with `python -m curla_amd.build` "
                            "(there is no CPU/PyTorch fallback for the learner path)")
    # an in-tree library must come from the sources next to it (a stale .so silently running old kernels is the worst
    # failure mode of an in-place build).  CURLA_SKIP_SRCHASH=1 turns the check off for a library built some other
    # way (a C-ABI consumer's own build, an install without the sources).
    if LIB_PATH == os.path.join(_HERE, "libcurla_hip.so") and os.environ.get("CURLA_SKIP_SRCHASH", "0") != "1":
        from . import build
        try:
            want = build.source_hash()
        except OSError as e:
            raise CurlaHipError(f"cannot check {LIB_PATH} against its sources ({e}); set CURLA_SKIP_SRCHASH=1 to load "
                                "a library that was built elsewhere") from e
        if build.built_hash() != want:
            raise CurlaHipError(f"{LIB_PATH} was built from other sources than the ones in curla_amd/csrc "
                                f"(stamp {build.built_hash()} != {want}): rebuild it with "
                                "`python -m curla_amd.build` (or CURLA_SKIP_SRCHASH=1 for an externally built library)")
    lib = ctypes.CDLL(LIB_PATH)
    # the argument lists below are positional: a library with another ABI number would be called with shifted
    # arguments (silent memory corruption, not an error) -- this matters most for CURLA_LIB_PATH builds, which skip
    # the source-stamp check above
    try:
        lib.curla_abi_version.restype = c_int
        got = int(lib.curla_abi_version())
    except AttributeError:
        got = None
    if got != ABI_VERSION:
        raise CurlaHipError(f"{LIB_PATH} has C-ABI version {got}, this binding was written for {ABI_VERSION} "
                            "(include/curla_hip.h CURLA_ABI_VERSION): rebuild the library or use the matching package")
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library drift apart
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, c_int)
    _lib = lib
    return lib


_trace_hook = None


def set_trace_hook(fn):
    """Host-logic tests only: route every kernel call to ``fn(name, args)`` instead
    of the library, to inspect the launch schedule on a machine without a GPU.
    Nothing is computed while a hook is installed.  Pass None to remove it."""
    global _trace_hook
    _trace_hook = fn


def call(name, *args):
    """Invoke an int-returning entry point; raise on a non-zero status."""
    if _trace_hook is not None:
        _trace_hook(name, args)
        return
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise CurlaHipError(f"{name} failed: {_ERRORS.get(rc, rc)}")


OPTIONS = ("conv1_u8", "conv1_f32", "s1_fwd", "bwd_split", "gemm_tile", "linear_bwd", "gemm_mfma", "s1_wgrad", "wgrad1_u8")  # curla_amd/csrc/options.h


def set_option(name, value):
    """curla_set_option: choose a kernel variant at run time (include/curla_hip.h lists names and values).  Runs on
    the host only (no GPU needed); raises on an unknown name or value."""
    rc = load().curla_set_option(str(name).encode(), str(value).encode())
    if rc != 0:
        raise CurlaHipError(f"curla_set_option({name!r}, {value!r}): unknown option or value")


def get_option(name):
    v = load().curla_get_option(str(name).encode())
    if v is None:
        raise CurlaHipError(f"curla_get_option({name!r}): unknown option")
    return v.decode()


class option:
    """``with _lib.option("conv1_u8", "rw"): ...`` -- a variant for the duration of a block (tests, A/B timing)."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    if _trace_hook is not None:
        return 0
    import torch
    return torch.cuda.current_stream().cuda_stream
