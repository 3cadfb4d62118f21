"""ctypes binding of libntm.so (the C ABI in include/ntm.h).

There is deliberately no fallback of any kind: if the HIP library is missing or no HIP device is
present, calls raise.  Nothing here imports or calls oracle/.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NTM_LIB_PATH") or os.path.join(_HERE, "libntm.so")   # override: kernel A/B builds
LAB_PATH = os.environ.get("NTM_LAB_PATH") or os.path.join(_HERE, "libntm_lab.so")   # laboratory kernels + diagnostics

NTM_GRU_AUTO, NTM_GRU_MFMA, NTM_GRU_VALU, NTM_GRU_MFMA2, NTM_GRU_F16X3, NTM_GRU_MFMA3, NTM_GRU_LAT, NTM_GRU_MFMA4, NTM_GRU_BF16X3 = 0, 1, 2, 3, 4, 5, 6, 7, 8
VARIANTS = {"auto": NTM_GRU_AUTO, "mfma": NTM_GRU_MFMA, "valu": NTM_GRU_VALU, "mfma2": NTM_GRU_MFMA2,
            "f16x3": NTM_GRU_F16X3, "lat": NTM_GRU_LAT, "bf16x3": NTM_GRU_BF16X3}

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int

_SIGNATURES = {
    "ntm_abi_version": (_int, []),
    "ntm_last_error": (ctypes.c_char_p, []),
    "ntm_gru_forward": (_int, [_vp] * 6 + [_int, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "ntm_gru_forward_ex": (_int, [_vp] * 6 + [_int, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _vp]),
    "ntm_gru_forward_io": (_int, [_vp] * 6 + [_int, _int, _int, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "ntm_gru_forward_esr": (_int, [_vp] * 6 + [_int, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp]),
    "ntm_gru_forward_losses": (_int, [_vp] * 6 + [_int, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _i64, _vp, ctypes.c_float, _vp, _vp]),
    "ntm_delay_forward": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _int, _int, _vp, _vp]),
    "ntm_diffdel_gru_forward": (_int, [_vp] * 5 + [_int, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _int, _int,
                                                   _vp, _vp]),
    "ntm_diffdel_gru_forward_ex": (_int, [_vp] * 5 + [_int, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _int, _int,
                                                      _vp, _int, _vp]),
    "ntm_diffdel_gru_forward_esr": (_int, [_vp] * 5 + [_int, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _int, _vp, _vp, _i64, _vp, _vp]),
    "ntm_diffdel_gru_forward_losses": (_int, [_vp] * 5 + [_int, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _int, _vp, _vp, _i64, _vp,
                                              ctypes.c_float, _vp, _vp]),
    "ntm_esr_sums": (_int, [_vp, _vp, _i64, _i64, _i64, _int, _vp, _vp]),
    "ntm_esr_splits": (_int, [_i64, _i64, _i64]),
    "ntm_esr_dcpre_sums": (_int, [_vp, _vp, _i64, _i64, _i64, ctypes.c_float, _vp, _vp]),
    "ntm_spec_sums": (_int, [_vp, _vp, _i64, _i64, _i64, _int, _int, _int, ctypes.c_float, _int, _vp, _vp]),
    "ntm_stft_sums": (_int, [_vp, _vp, _i64, _i64, _i64, _int, _int, _int, ctypes.c_float, _int, _vp, _vp]),
    "ntm_mel_sums": (_int, [_vp, _vp, _i64, _i64, _i64, _int, _int, _int, ctypes.c_float, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "ntm_copy2d_async": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _int, _vp]),
    "ntm_demodulate": (_int, [_vp, _vp, _int, _i64, _vp, _int, _i64, _i64, _vp, _vp]),
    "ntm_tape_record_field": (_int, [_vp, _vp, _vp, _i64, _i64, ctypes.c_double, ctypes.c_double, _vp]),
    "ntm_tape_hmag": (_int, [_vp, _vp, _i64, _i64, _vp, ctypes.c_double, ctypes.POINTER(ctypes.c_double), _vp]),
    "ntm_resample_fir": (_int, [_vp, _vp, _i64, _i64, _i64, _int, _int, _int, _vp, _vp]),
    "ntm_fir_f64": (_int, [_vp, _vp, _i64, _i64, _vp, _int, _int, _vp]),
    "ntm_tcn_forward": (_int, [_vp, _int, _int, _int, ctypes.POINTER(_int), _vp, _vp, _i64, _i64, _vp, _vp]),
    "ntm_tcn_scratch_floats": (_i64, [_i64, _i64, _int]),
    "ntm_tcn_chunk_streams": (_i64, [_i64, _i64, _int]),
    "ntm_loss_scalars": (_int, [_vp, _i64, _i64, ctypes.c_double, _vp, _vp]),
}

# include/ntm_lab.h: libntm_lab.so (older / experimental GRU kernels, diagnostic builds) -- tests and tools only
_LAB_SIGNATURES = {
    "ntm_lab_last_error": (ctypes.c_char_p, []),
    "ntm_lab_gru_forward": (_int, [_vp] * 6 + [_int, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _vp]),
    "ntm_debug_gru_stamps": (_int, [_vp] * 6 + [_vp, _vp, _i64, _i64, _vp, _vp, _int, _vp]),
    "ntm_debug_transpose4": (_int, [_vp, _vp, _vp]),
    "ntm_debug_gru_ablate": (_int, [_vp] * 6 + [_vp, _vp, _i64, _i64, _vp, _int, _vp]),
    "ntm_lab_tcn_forward": (_int, [_vp, _int, _int, _int, _vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "ntm_lab_tcn_stamps": (_int, [_vp]),
    "ntm_lab_tcn_trace": (_int, [_vp]),
}
LAB_VARIANTS = ("mfma", "valu")

_lib = None
_lab = None
ABI_VERSION = 9          # include/ntm.h NTM_ABI_VERSION this binding was written against
HIDDEN_SIZES = (8, 16, 32, 64)      # sizes with a kernel of their own; every H in [1, MAX_HIDDEN] runs (include/ntm.h)
MAX_HIDDEN = 1024
NTM_DIFFDEL_AUTO, NTM_DIFFDEL_TWO_PASS, NTM_DIFFDEL_FUSED = 0, 1, 2
DIFFDEL_MODES = {"auto": NTM_DIFFDEL_AUTO, "two_pass": NTM_DIFFDEL_TWO_PASS, "fused": NTM_DIFFDEL_FUSED}


class NtmError(RuntimeError):
    """A libntm.so entry point returned a negative status."""


def build(verbose=False):
    """Compile libntm.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", os.path.join(_HERE, "csrc")] + ([] if verbose else ["-s"]), check=True)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NtmError(f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                           "(there is no CPU fallback for this path)")
        handle = ctypes.CDLL(LIB_PATH)
        # the version first: a stale library (the .so files are build artefacts) lacks newer symbols, and the helpful
        # message must come before the bare AttributeError of a missing one
        try:
            ver = handle.ntm_abi_version
        except AttributeError:
            raise NtmError(f"{LIB_PATH} does not export ntm_abi_version: not a libntm.so of this tree, rebuild it") from None
        ver.restype, ver.argtypes = _SIGNATURES["ntm_abi_version"]
        if ver() != ABI_VERSION:
            raise NtmError(f"{LIB_PATH} has ABI version {ver()}, this binding needs {ABI_VERSION}: rebuild it "
                           f"(`make -C {os.path.join(_HERE, 'csrc')}`)")
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def lab():
    """libntm_lab.so: the laboratory kernels (kernel_variant in LAB_VARIANTS) and the diagnostic entry points.  The
    product path (kernel_variant "auto" / "mfma2" / "lat" / "f16x3" / "bf16x3") never loads it."""
    global _lab
    if _lab is None:
        if not os.path.exists(LAB_PATH):
            raise NtmError(f"{LAB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}`")
        handle = ctypes.CDLL(LAB_PATH)
        for name, (res, args) in _LAB_SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lab = handle
    return _lab


def gru_forward_fn(variant_name, hidden_size=64):
    """-> (entry point with ntm_gru_forward_ex's signature, error-message getter) for a kernel_variant name."""
    if variant_name in LAB_VARIANTS and hidden_size == 64:
        return lab().ntm_lab_gru_forward, lab().ntm_lab_last_error
    return lib().ntm_gru_forward_ex, lib().ntm_last_error


def check(rc, what, last_error=None):
    if rc != 0:
        raise NtmError(f"{what} failed ({rc}): {(last_error or lib().ntm_last_error)().decode()}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
