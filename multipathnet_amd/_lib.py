"""ctypes binding of libmpn_hip.so — the ONLY compute backend of this package.

There is no CPU fallback: if the HIP library is missing, or no gfx950 device is usable, every op
raises.  (The CPU restatement under oracle/ is test infrastructure and is never imported here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmpn_hip.so")
_DBG_PATH = os.path.join(_HERE, "libmpn_hip_dbg.so")  # same sources, -DMPN_DEBUG_HOOKS: mpn_debug_* test / timing hooks
_libs = {}
_flavour = "debug" if os.environ.get("MPN_FLAVOUR") == "debug" else "product"

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)


class MpnError(RuntimeError):
    pass


class FrcnnConfig(C.Structure):
    """mirror of mpn_frcnn_config (include/mpn.h)"""
    _fields_ = [
        ("n_conv", C.c_int), ("conv_cout", C.POINTER(C.c_int)), ("pool_after", C.POINTER(C.c_int)),
        ("pooled_h", C.c_int), ("pooled_w", C.c_int), ("spatial_scale", C.c_float), ("fc_dim", C.c_int),
        ("n_classes", C.c_int), ("max_h", C.c_int), ("max_w", C.c_int), ("max_rois", C.c_int),
        ("tf_scale", C.c_double), ("tf_mean", C.c_double * 3), ("tf_std", C.c_double * 3), ("tf_swap", C.c_int * 3),
        ("bbox_mean", C.c_float * 4), ("bbox_std", C.c_float * 4), ("nms_thresh", C.c_float),
        ("score_thresh", C.c_float), ("top_k", C.c_int),
        ("num_iter", C.c_int), ("bbox_voting", C.c_int), ("bbox_vote_thresh", C.c_float), ("bbox_vote_score_pow", C.c_float),
        ("scale_target", C.c_double), ("scale_max", C.c_double), ("use_rbox_scores", C.c_int), ("roi_bin_rule", C.c_int), ("fc_arith", C.c_int),
    ]


class MpnetWeights(C.Structure):
    """mirror of mpn_mpnet_weights (include/mpn.h)"""
    _fields_ = [
        ("n_towers", C.c_int), ("region", C.c_int * 8), ("use_conv4", C.c_int * 8), ("use_conv3", C.c_int * 8),
        ("tap_conv3", C.c_int), ("tap_conv4", C.c_int), ("n_integral", C.c_int),
        ("mix_w", f32p * 8), ("mix_b", f32p * 8), ("fc6_w", f32p * 8), ("fc6_b", f32p * 8), ("fc7_w", f32p * 8), ("fc7_b", f32p * 8),
        ("conv345_unnormalized", C.c_int),
    ]


class ResnetWeights(C.Structure):
    """mirror of mpn_resnet_weights (include/mpn.h)"""
    _fields_ = [
        ("n_convs", C.c_int), ("w", C.POINTER(f32p)), ("b", C.POINTER(f32p)),
        ("cin", C.POINTER(C.c_int)), ("cout", C.POINTER(C.c_int)), ("ksize", C.POINTER(C.c_int)), ("stride", C.POINTER(C.c_int)),
        ("pad", C.POINTER(C.c_int)),
        ("n_blocks", C.c_int), ("block_n_convs", C.POINTER(C.c_int)), ("block_has_shortcut", C.POINTER(C.c_int)),
        ("n_trunk_blocks", C.c_int),
        ("n_heads", C.c_int), ("head_region", C.c_int * 8), ("n_integral", C.c_int), ("bf16", C.c_int),
    ]


class GraphOp(C.Structure):
    """mirror of mpn_graph_op (include/mpn.h)"""
    _fields_ = [("kind", C.c_int), ("src", C.c_int), ("dst", C.c_int), ("dst_c_off", C.c_int), ("cin", C.c_int), ("cout", C.c_int),
                ("kh", C.c_int), ("kw", C.c_int), ("sh", C.c_int), ("sw", C.c_int), ("ph", C.c_int), ("pw", C.c_int), ("relu", C.c_int),
                ("w", f32p), ("b", f32p), ("src_c_off", C.c_int), ("ceil_mode", C.c_int), ("lrn_alpha", C.c_float), ("lrn_beta", C.c_float),
                ("lrn_k", C.c_float)]


class GraphWeights(C.Structure):
    """mirror of mpn_graph_weights (include/mpn.h)"""
    _fields_ = [("n_trunk_ops", C.c_int), ("trunk_ops", C.POINTER(GraphOp)), ("n_trunk_tensors", C.c_int), ("trunk_tensor_c", C.POINTER(C.c_int)),
                ("feat_tensor", C.c_int),
                ("n_head_ops", C.c_int), ("head_ops", C.POINTER(GraphOp)), ("n_head_tensors", C.c_int), ("head_tensor_c", C.POINTER(C.c_int)),
                ("out_tensor", C.c_int), ("bf16", C.c_int), ("n_heads", C.c_int), ("head_region", C.c_int * 8), ("n_integral", C.c_int)]


def lib_path():
    return _LIB_PATH


def load(flavour=None):
    """Load libmpn_hip.so (built in-tree by `make -C multipathnet_amd/csrc` / __graft_entry__.build()).

    flavour None = the process's current one: "product" unless MPN_FLAVOUR=debug or inside `debug_hooks()`.  The debug
    flavour (libmpn_hip_dbg.so) is the same code with the mpn_debug_* variant / ablation hooks compiled in; it exists for
    the test suite and tools/ only.  The two libraries share no state: a handle must be used with the library that made it
    (models.FastRCNN remembers its own)."""
    fl = flavour or _flavour
    if fl in _libs:
        return _libs[fl]
    path = _DBG_PATH if fl == "debug" else _LIB_PATH
    if not os.path.exists(path):
        raise MpnError(
            "%s not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). multipathnet_amd has no CPU fallback." % (os.path.basename(path), path))
    # One HIP runtime per process: PyTorch ships its own libamdhip64 (SONAME libamdhip64.so.7).  Loaded first, it is the one
    # this library's DT_NEEDED resolves to; loaded second, the process would hold two runtimes with separate device state
    # (torch's tensors would be foreign pointers here).  A non-Python host has a single system runtime and no such issue.
    import torch  # noqa: F401
    lib = C.CDLL(path, mode=C.RTLD_LOCAL)
    lib.mpn_last_error.restype = C.c_char_p
    lib.mpn_pick_scale.restype = C.c_double
    lib.mpn_pick_scale.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double]
    lib.mpn_conv3x3_workspace_bytes.restype = C.c_size_t
    lib.mpn_det_record_floats.restype = C.c_size_t
    lib.mpn_frcnn_shard_rows_floats.restype = C.c_size_t
    lib.mpn_frcnn_shard_class_floats.restype = C.c_size_t
    lib.mpn_gather_rows.argtypes = [C.c_void_p, f32p, C.c_size_t, f32p, C.c_void_p]
    lib.mpn_comm_destroy.restype = None
    lib.mpn_frcnn_destroy.restype = None
    _libs[fl] = lib
    return lib


class debug_hooks(object):
    """`with _lib.debug_hooks() as lib:` — inside the block every op of this package runs on libmpn_hip_dbg.so (the flavour
    with the mpn_debug_* hooks); `lib` is that library.  Test / tools use only."""

    def __enter__(self):
        global _flavour
        self._prev = _flavour
        _flavour = "debug"
        return load("debug")

    def __exit__(self, *exc):
        global _flavour
        _flavour = self._prev
        return False


def check(rc, what=""):
    if rc != 0:
        raise MpnError("%s failed (status %d): %s" % (what or "mpn call", rc, load().mpn_last_error().decode()))


def require_gpu():
    """Fail loudly unless a HIP device is usable (no silent fallback)."""
    import torch
    if not torch.cuda.is_available():
        raise MpnError("multipathnet_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU path")
