"""ctypes binding of libavvad_hip.so (the C ABI declared in include/avvad.h).

There is NO fallback: if the shared library is missing or a call returns a
negative AVVAD_E* code, an exception is raised.  The product path never runs
on anything but the HIP kernels.
"""
import ctypes as C
import os

# PyTorch must be imported BEFORE the library is dlopen'ed: torch ships its own libamdhip64 and the kernels
# have to run on THAT runtime instance (same device context, streams and allocations).  Loading
# libavvad_hip.so first would pull a second HIP runtime into the process ("no ROCm-capable device").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AVVAD_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libavvad_hip.so")

TRUNK_NCONV = 20
ABI_VERSION = 3          # include/avvad.h AVVAD_ABI_VERSION: the signatures below describe exactly this version
_ERR = {-1: "AVVAD_EINVAL (bad descriptor / unsupported shape)", -2: "AVVAD_EWORKSPACE (workspace too small)",
        -3: "AVVAD_ELAUNCH (kernel launch failed)"}

FP = C.c_void_p  # device pointers travel as integers


class AvvadError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("M", "N", "K", "lda", "ldb", "ldc", "transA", "transB", "accumulate",
                                       "split_k", "relu_a", "relu_b")]


class WavenetDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("L", C.c_int), ("qc", C.c_int), ("R", C.c_int), ("D", C.c_int), ("Bn", C.c_int),
                ("fw", C.c_int), ("P", C.c_int), ("n_layers", C.c_int), ("dilations_h", C.POINTER(C.c_int)),
                ("use_bias", C.c_int), ("save_for_backward", C.c_int), ("shared_device", C.c_int)]


class WavenetPtrs(C.Structure):  # avvad_wavenet_params and avvad_wavenet_grads share this layout
    _fields_ = [("causal_w", FP), ("causal_b", FP), ("dil_w_h", C.POINTER(FP)), ("dil_b_h", C.POINTER(FP)),
                ("dense_w_h", C.POINTER(FP)), ("dense_b_h", C.POINTER(FP)), ("bott_w", FP), ("bott_b", FP)]


class TrunkDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("training", C.c_int), ("momentum", C.c_float),
                ("eps", C.c_float), ("save_for_backward", C.c_int)]


class TrunkParams(C.Structure):
    _fields_ = [("conv_w", FP * TRUNK_NCONV), ("bn_w", FP * TRUNK_NCONV), ("bn_b", FP * TRUNK_NCONV),
                ("bn_rm", FP * TRUNK_NCONV), ("bn_rv", FP * TRUNK_NCONV)]


class TrunkGrads(C.Structure):
    _fields_ = [("conv_w", FP * TRUNK_NCONV), ("bn_w", FP * TRUNK_NCONV), ("bn_b", FP * TRUNK_NCONV)]


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "H", "W", "C", "Co", "KS", "stride", "pad")]


class McbDesc(C.Structure):
    _fields_ = [("rows", C.c_int), ("A", C.c_int), ("V", C.c_int), ("D", C.c_int), ("eps", C.c_float),
                ("training", C.c_int), ("momentum", C.c_float), ("save_for_backward", C.c_int)]


class StftDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("L", C.c_long), ("n_fft", C.c_int), ("hop", C.c_int), ("T", C.c_int), ("eps", C.c_float)]


class LstmDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("In", C.c_int), ("H", C.c_int), ("lengths", FP),
                ("save_for_backward", C.c_int)]


# name -> (restype, argtypes); the list doubles as the symbol inventory checked by the CPU tests
SIGNATURES = {
    "avvad_version": (C.c_char_p, []),
    "avvad_abi_version": (C.c_int, []),
    "avvad_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "avvad_get_option": (C.c_int, [C.c_char_p]),
    "avvad_engine_workspace": (C.c_size_t, []),
    "avvad_gemm_f32": (C.c_int, [FP, FP, FP, FP, C.POINTER(GemmDesc), FP, C.c_size_t, FP]),
    "avvad_wavenet_workspace": (C.c_size_t, [C.POINTER(WavenetDesc)]),
    "avvad_wavenet_fwd": (C.c_int, [FP, C.POINTER(WavenetPtrs), FP, C.POINTER(WavenetDesc), FP, C.c_size_t, FP]),
    "avvad_wavenet_bwd": (C.c_int, [FP, C.POINTER(WavenetPtrs), FP, C.POINTER(WavenetPtrs), FP,
                                    C.POINTER(WavenetDesc), FP, C.c_size_t, FP]),
    "avvad_wavenet_block_fwd": (C.c_int, [FP, FP, FP, FP, FP, FP, C.c_int, C.c_int, C.c_int, FP]),
    "avvad_conv2d_pack_weights": (C.c_int, [FP, FP, FP, C.POINTER(ConvDesc), FP]),
    "avvad_conv2d_fwd": (C.c_int, [FP, FP, FP, C.POINTER(ConvDesc), FP, C.c_size_t, FP]),
    "avvad_conv2d_dgrad": (C.c_int, [FP, FP, FP, C.POINTER(ConvDesc), C.c_int, FP, C.c_size_t, FP]),
    "avvad_conv2d_wgrad": (C.c_int, [FP, FP, FP, C.POINTER(ConvDesc), FP, C.c_size_t, FP]),
    "avvad_conv2d_pack_weights_bf16": (C.c_int, [FP, FP, FP, C.POINTER(ConvDesc), FP]),
    "avvad_conv2d_fwd_bf16": (C.c_int, [FP, FP, FP, C.POINTER(ConvDesc), FP, C.c_size_t, FP]),
    "avvad_conv2d_dgrad_bf16": (C.c_int, [FP, FP, FP, C.POINTER(ConvDesc), C.c_int, FP, C.c_size_t, FP]),
    "avvad_conv2d_wgrad_bf16": (C.c_int, [FP, FP, FP, C.POINTER(ConvDesc), FP, C.c_size_t, FP]),
    "avvad_trunk_workspace": (C.c_size_t, [C.POINTER(TrunkDesc)]),
    "avvad_trunk_activation": (C.c_int, [C.POINTER(TrunkDesc), C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                         C.POINTER(C.c_int)]),
    "avvad_trunk_fwd": (C.c_int, [FP, C.POINTER(TrunkParams), FP, C.POINTER(TrunkDesc), FP, C.c_size_t, FP]),
    "avvad_trunk_bwd": (C.c_int, [FP, C.POINTER(TrunkParams), FP, C.POINTER(TrunkGrads), C.POINTER(TrunkDesc), FP,
                                  C.c_size_t, FP]),
    "avvad_lstm_workspace": (C.c_size_t, [C.POINTER(LstmDesc)]),
    "avvad_lstm_layer_fwd": (C.c_int, [FP, FP, FP, FP, FP, FP, C.POINTER(LstmDesc), FP, C.c_size_t, FP]),
    "avvad_lstm_layer_bwd": (C.c_int, [FP, FP, FP, FP, FP, FP, FP, FP, FP, FP, C.POINTER(LstmDesc), FP, C.c_size_t, FP]),
    "avvad_mcb_workspace": (C.c_size_t, [C.POINTER(McbDesc)]),
    "avvad_mcb_fusion_fwd": (C.c_int, [FP] * 11 + [C.POINTER(McbDesc), FP, C.c_size_t, FP]),
    "avvad_mcb_fusion_bwd": (C.c_int, [FP] * 12 + [C.POINTER(McbDesc), FP, C.c_size_t, FP]),
    "avvad_count_sketch_fwd": (C.c_int, [FP, FP, FP, FP, C.c_int, C.c_int, C.c_int, FP]),
    "avvad_count_sketch_bwd": (C.c_int, [FP, FP, FP, FP, C.c_int, C.c_int, C.c_int, FP]),
    "avvad_mcb_fwd": (C.c_int, [FP] * 7 + [C.c_int] * 4 + [FP]),
    "avvad_mcb_bwd": (C.c_int, [FP] * 9 + [C.c_int] * 4 + [FP]),
    "avvad_stft_workspace": (C.c_size_t, [C.POINTER(StftDesc)]),
    "avvad_stft": (C.c_int, [FP, FP, C.POINTER(StftDesc), C.c_int, FP, C.c_size_t, FP]),
    "avvad_stft_features": (C.c_int, [FP, FP, FP, FP, C.POINTER(StftDesc), C.c_float, FP, C.c_size_t, FP]),
    "avvad_peak_normalize": (C.c_int, [FP, FP, C.c_int, C.c_long, FP]),
    "avvad_standardize": (C.c_int, [FP, FP, FP, FP, C.c_size_t, C.c_int, C.c_int, C.c_float, FP]),
    "avvad_bce_2classes": (C.c_int, [FP, FP, FP, FP, FP, FP, C.c_long, C.c_int, C.c_float, FP]),
    "avvad_bce_masked": (C.c_int, [FP, FP, FP, FP, FP, C.c_int, C.c_int, C.c_int, C.c_float, FP]),
    "avvad_adam_step": (C.c_int, [FP, FP, FP, FP, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, FP]),
    "avvad_copy_cols": (C.c_int, [FP, FP, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, FP]),
    "avvad_colsum_acc": (C.c_int, [FP, C.c_size_t, C.c_int, FP, FP]),
    "avvad_scale_by_device_scalar": (C.c_int, [FP, FP, C.c_size_t, FP]),
    "avvad_transpose_last2": (C.c_int, [FP, FP, C.c_int, C.c_int, C.c_int, FP]),
}

_lib = None


def lib():
    """Load (once) and return the shared library; raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AvvadError("%s not found: build it with `python __graft_entry__.py` (or csrc/build.sh); "
                             "there is no CPU/PyTorch fallback for the AV-VAD hot path" % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        got = handle.avvad_abi_version()
        if got != ABI_VERSION:       # a stale build would read e.g. the workspace pointer as the stream and fault the GPU
            raise AvvadError("%s has ABI version %d, this binding expects %d: rebuild it (python __graft_entry__.py)"
                             % (LIB_PATH, got, ABI_VERSION))
        _lib = handle
    return _lib


def set_option(name, value):
    """Process-wide schedule option of the library (include/avvad.h: avvad_set_option)."""
    check(lib().avvad_set_option(name.encode(), int(value)), "avvad_set_option(%s)" % name)


def get_option(name):
    return lib().avvad_get_option(name.encode())


def check(rc, what):
    if rc != 0:
        raise AvvadError("%s failed: %s" % (what, _ERR.get(rc, rc)))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def ptr_array(ts):
    arr = (FP * max(1, len(ts)))()
    for i, t in enumerate(ts):
        arr[i] = None if t is None else t.data_ptr()
    return arr
