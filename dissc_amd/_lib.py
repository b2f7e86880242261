"""ctypes binding of libdissc_hip.so (C ABI: include/dissc_hip.h).

torch is imported first on purpose: the library's DT_NEEDED libamdhip64.so.7 then
resolves to the HIP runtime PyTorch already loaded, so device pointers and
hipStream_t handles are shared between PyTorch (allocator, streams, RCCL) and the
kernels.  If the library is missing this module raises -- there is no fallback.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL load, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libdissc_hip.so"


def library_path():
    # DISSC_HIP_LIB: development hook for A/B-ing an alternative build of the same library
    return os.environ.get("DISSC_HIP_LIB") or os.path.join(_HERE, _LIB_NAME)


class DisscError(RuntimeError):
    pass


class DisscTensor(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.c_void_p),
                ("shape", ctypes.c_int64 * 4), ("ndim", ctypes.c_int32)]


MAX_UPS, MAX_RK = 8, 4


class DisscGenConfig(ctypes.Structure):
    _fields_ = [("model_in_dim", ctypes.c_int32), ("upsample_initial_channel", ctypes.c_int32),
                ("num_upsamples", ctypes.c_int32), ("upsample_rates", ctypes.c_int32 * MAX_UPS),
                ("upsample_kernel_sizes", ctypes.c_int32 * MAX_UPS), ("num_kernels", ctypes.c_int32),
                ("resblock_kernel_sizes", ctypes.c_int32 * MAX_RK),
                ("resblock_dilations", (ctypes.c_int32 * 3) * MAX_RK),
                ("num_embeddings", ctypes.c_int32), ("embedding_dim", ctypes.c_int32),
                ("num_speakers", ctypes.c_int32), ("has_f0", ctypes.c_int32), ("has_spkr", ctypes.c_int32)]


def _load():
    path = library_path()
    if not os.path.exists(path):
        raise DisscError(
            f"{path} not found: build the HIP library first "
            "(python -c 'import __graft_entry__ as g; g.build()').  "
            "dissc_amd has no CPU fallback.")
    L = ctypes.CDLL(path)
    vp, i32, i64p = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
    L.dissc_last_error.restype = ctypes.c_char_p
    L.dissc_abi_version.restype = i32
    L.dissc_device_count.restype = i32
    L.dissc_device_name.argtypes = [i32, ctypes.c_char_p, ctypes.c_size_t]
    L.dissc_gen_create.argtypes = [ctypes.POINTER(DisscGenConfig), ctypes.POINTER(DisscTensor),
                                   ctypes.c_size_t, ctypes.POINTER(vp)]
    L.dissc_gen_create_ex.argtypes = [ctypes.POINTER(DisscGenConfig), ctypes.POINTER(DisscTensor),
                                      ctypes.c_size_t, i32, ctypes.POINTER(vp)]
    L.dissc_gen_destroy.argtypes = [vp]
    L.dissc_gen_destroy.restype = None
    L.dissc_gen_hop.argtypes = [vp]
    L.dissc_gen_workspace_bytes.argtypes = [vp, i32, i32]
    L.dissc_gen_workspace_bytes.restype = ctypes.c_size_t
    L.dissc_gen_flops.argtypes = [vp, ctypes.c_int64]
    L.dissc_gen_flops.restype = ctypes.c_double
    L.dissc_gen_flops_executed.argtypes = [vp, ctypes.c_int64]
    L.dissc_gen_flops_executed.restype = ctypes.c_double
    L.dissc_gen_forward.argtypes = [vp, i64p, vp, i64p, vp, i32, i32, vp, vp, ctypes.c_size_t, vp]
    L.dissc_conv1d.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32,
                               ctypes.c_float, vp]
    L.dissc_conv_transpose1d.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32,
                                         ctypes.c_float, vp]
    L.dissc_pred_create.argtypes = [i32, ctypes.POINTER(DisscTensor), ctypes.c_size_t, ctypes.POINTER(vp)]
    L.dissc_pred_destroy.argtypes = [vp]
    L.dissc_pred_destroy.restype = None
    L.dissc_pred_workspace_bytes.argtypes = [vp, i32, i32]
    L.dissc_pred_workspace_bytes.restype = ctypes.c_size_t
    L.dissc_len_set_norm.argtypes = [vp, ctypes.c_float, ctypes.c_float]
    L.dissc_len_forward.argtypes = [vp, vp, vp, vp, i32, i32, vp, i32, vp, ctypes.c_size_t, vp]
    L.dissc_pitch_forward.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp, i32, vp, i32, vp,
                                      ctypes.c_size_t, vp]
    L.dissc_dedup.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp]
    L.dissc_len_carryover.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    L.dissc_expand.argtypes = [vp, vp, vp, i32, i32, vp, i32, vp]
    L.dissc_hubert_create.argtypes = [i32, ctypes.POINTER(DisscTensor), ctypes.c_size_t, vp, i32,
                                      ctypes.POINTER(vp)]
    L.dissc_hubert_destroy.argtypes = [vp]
    L.dissc_hubert_destroy.restype = None
    L.dissc_hubert_frames.argtypes = [i32]
    L.dissc_hubert_workspace_bytes.argtypes = [vp, i32, i32]
    L.dissc_hubert_workspace_bytes.restype = ctypes.c_size_t
    L.dissc_hubert_forward.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, ctypes.c_size_t, vp]
    L.dissc_kmeans_assign.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp]
    L.dissc_mfma_peak.argtypes = [i32, ctypes.POINTER(ctypes.c_float)]
    L.dissc_erf_check.argtypes = [vp, vp, i32, vp]
    L.dissc_set_option.argtypes = [ctypes.c_char_p, i32]
    L.dissc_get_option.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    L.dissc_conv_bench.argtypes = [i32] * 9 + [ctypes.POINTER(ctypes.c_float)]
    L.dissc_conv1d_s2.argtypes = [vp, vp, vp, vp, vp] + [i32] * 9 + [vp]
    L.dissc_conv_s2_bench.argtypes = [i32] * 5 + [ctypes.POINTER(ctypes.c_float)]
    L.dissc_pair_bench.argtypes = [i32] * 8 + [ctypes.POINTER(ctypes.c_float)]
    L.dissc_respair1d.argtypes = [vp] * 8 + [i32] * 6 + [ctypes.c_float, i32, ctypes.c_float, i32, vp]
    L.dissc_wav_postprocess.argtypes = [vp, vp, i32, i32, vp]
    L.dissc_pitch_stats.argtypes = [vp, vp, i32, vp, vp, vp, vp]
    i64 = ctypes.c_longlong
    L.dissc_pack_rows.argtypes = [vp, i64, vp, vp, i32, i32, vp, vp]
    L.dissc_comm_unique_id.argtypes = [ctypes.c_char_p]
    L.dissc_comm_create.argtypes = [ctypes.c_char_p, i32, i32, ctypes.POINTER(vp)]
    L.dissc_comm_destroy.argtypes = [vp]
    L.dissc_allgather_waves.argtypes = [vp, vp, ctypes.c_size_t, vp, vp]
    L.dissc_resample.argtypes = [vp, i32, vp, i32, ctypes.c_double, vp, vp, i32, i32, vp]
    return L


lib = _load()

# tuning hook: DISSC_OPTIONS="key=value,key=value" -> dissc_set_option (see include/dissc_hip.h)
for _kv in filter(None, os.environ.get("DISSC_OPTIONS", "").split(",")):
    _k, _v = _kv.split("=")
    if lib.dissc_set_option(_k.strip().encode(), int(_v)) != 0:
        raise DisscError(f"DISSC_OPTIONS: unknown option {_k}")


def check(rc, what=""):
    if rc != 0:
        msg = lib.dissc_last_error()
        raise DisscError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def current_stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def make_tensor_table(named):
    """{name: contiguous fp32 CPU tensor} -> (ctypes array, keep-alive list)."""
    arr = (DisscTensor * len(named))()
    keep = []
    for i, (k, t) in enumerate(named.items()):
        t = t.detach().to("cpu", torch.float32).contiguous()
        nb = k.encode()
        keep.append((t, nb))
        arr[i].name = nb
        arr[i].data = t.data_ptr()
        arr[i].ndim = t.dim()
        for d in range(t.dim()):
            arr[i].shape[d] = t.shape[d]
    return arr, keep
