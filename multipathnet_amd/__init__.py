"""multipathnet_amd — MI355X-native per-image detection hot path of MultiPathNet / Fast R-CNN.

The package is a thin host layer over libmpn_hip.so (hand-written gfx950 HIP kernels behind the C ABI
in include/mpn.h).  It mirrors the reference's operator surface for the hot path only:

    nn      inn.ROIPooling, nn.Foveal, nn.ContextRegion, nn.BBoxNorm, nn.SelectBoxes, ImageTransformer, ...
    utils   utils.nms / bbox_vote / boxoverlap / convertFrom / keep_top_k
    models  FastRCNN (models/vgg.lua graph) as one fused device pipeline
    detect  ImageDetect, Tester_FRCNN

No CPU fallback exists: importing works anywhere, running an op without the HIP library or a
device raises MpnError.
"""
from . import _lib  # noqa: F401
from ._lib import MpnError, lib_path, load  # noqa: F401

__all__ = ["MpnError", "lib_path", "load"]
