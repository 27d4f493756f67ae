"""Host-side mirror of the reference's nn.Module surface for the hot path.

Same names, argument meaning and error behaviour as the Lua modules (`:updateOutput(input)` /
`:forward(input)`, `self.output`), so the parity tests read like the reference's own tests.  Every
`updateOutput` is a thin call through the C ABI of libmpn_hip.so (include/mpn.h) on raw device
pointers of torch CUDA tensors; torch is used for device memory and streams only.

Reference classes mirrored (file:line relative to /root/reference):
  inn.ROIPooling            — external `inn` rock; call sites models/vgg.lua:28, alexnet.lua:23, ...
  nn.Foveal                 — modules/Foveal.lua:9-44
  nn.ContextRegion          — modules/ContextRegion.lua:9-32
  nn.BBoxNorm               — modules/BBoxNorm.lua:9-32
  nn.SelectBoxes            — modules/SelectBoxes.lua:9-56
  fbcoco.ImageTransformer   — modules/ImageTransformer.lua:9-33
  nn.SoftMax / nn.Linear / cudnn.SpatialConvolution(3x3) / nn.SpatialMaxPooling(2,2,2,2):ceil() / nn.ReLU
                            — external nn/cudnn rocks, used by models/vgg.lua:14-31
  nn.ModeSwitch             — modules/ModeSwitch.lua (graph semantics only: evaluate -> branch 2)
"""
import ctypes as C

import torch

from . import _lib
from ._lib import f32p, i32p, check


def _f(t, name="tensor"):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise _lib.MpnError("%s must live on the HIP device (got a CPU tensor); multipathnet_amd has no CPU path" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32 (torch.FloatTensor semantics)" % name)
    assert t.is_contiguous(), "%s must be contiguous" % name
    return C.cast(t.data_ptr(), f32p)


def _i(t):
    return C.cast(t.data_ptr(), i32p)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Module(object):
    def __init__(self):
        self.output = None
        self.train = False  # hot path = evaluate mode (test_runner.lua:37,43)

    def forward(self, input):
        return self.updateOutput(input)

    __call__ = forward

    def evaluate(self):
        self.train = False
        return self

    def training(self):
        self.train = True
        return self


class ROIPooling(Module):
    """inn.ROIPooling(W, H, spatial_scale): input {features [B,C,h,w], rois [N,5]} -> [N,C,H,W].

    `coord_offset` / `end_adjust` parameterise the unpinned coordinate convention;
    defaults are the README.md:202-203 "v2" behaviour.  `bin_rule`: BINS_CAFFE = the module's CUDA branch (default),
    BINS_ADAPTIVE = its CPU branch, crop + nn.SpatialAdaptiveMaxPooling (include/mpn.h MPN_ROI_BINS_*)."""
    BINS_CAFFE, BINS_ADAPTIVE = 0, 1

    def __init__(self, W, H, spatial_scale=1.0, coord_offset=1.0, end_adjust=0, bin_rule=0):
        super().__init__()
        self.W, self.H, self.spatial_scale = int(W), int(H), float(spatial_scale)
        self.coord_offset, self.end_adjust, self.bin_rule = float(coord_offset), int(end_adjust), int(bin_rule)
        self.indices = None

    def setSpatialScale(self, scale):
        self.spatial_scale = float(scale)
        return self

    def updateOutput(self, input):
        feat, rois = input
        assert feat.dim() == 4 and rois.dim() == 2 and rois.size(1) == 5
        B, Cc, h, w = feat.shape
        N = rois.size(0)
        self.output = torch.empty((N, Cc, self.H, self.W), dtype=torch.float32, device=feat.device)
        self.indices = torch.empty((N, Cc, self.H, self.W), dtype=torch.int32, device=feat.device)
        if N:
            check(_lib.load().mpn_roi_pool_forward_rule(_f(feat, "features"), B, Cc, h, w, _f(rois, "rois"), N, self.H, self.W,
                                                        C.c_float(self.spatial_scale), C.c_float(self.coord_offset),
                                                        self.end_adjust, self.bin_rule, _f(self.output), _i(self.indices), _stream()),
                  "ROIPooling")
        return self.output


class Foveal(Module):
    def updateOutput(self, input):
        assert input.dim() == 2
        assert input.size(1) == 5
        N = input.size(0)
        self.output = torch.empty((N * 4, 5), dtype=torch.float32, device=input.device)
        if N:
            check(_lib.load().mpn_foveal_forward(_f(input), N, _f(self.output), _stream()), "Foveal")
        return self.output


class ContextRegion(Module):
    def __init__(self, scale):
        super().__init__()
        self.scale = float(scale)

    def updateOutput(self, input):
        assert input.dim() == 2
        assert input.size(1) == 5
        self.output = torch.empty_like(input)
        if input.size(0):
            check(_lib.load().mpn_context_region_forward(_f(input), input.size(0), C.c_double(self.scale), _f(self.output),
                                                         _stream()), "ContextRegion")
        return self.output


class BBoxNorm(Module):
    def __init__(self, mean, std):
        assert mean is not None and std is not None
        super().__init__()
        self.mean = [float(v) for v in mean]
        self.std = [float(v) for v in std]

    def updateOutput(self, input):
        assert input.dim() == 2 and input.size(1) % 4 == 0
        self.output = input
        if not self.train:
            if not input.is_contiguous():
                self.output = input.contiguous()
            m = (C.c_float * 4)(*self.mean)
            s = (C.c_float * 4)(*self.std)
            check(_lib.load().mpn_bbox_norm_forward(_f(self.output), self.output.size(0), self.output.size(1), m, s, _stream()),
                  "BBoxNorm")
        return self.output


class SelectBoxes(Module):
    def updateOutput(self, input):
        classes, ys = input
        B = classes.size(0)
        self.output = torch.empty((B, 4), dtype=torch.float32, device=classes.device)
        if B:
            check(_lib.load().mpn_select_boxes_forward(_f(classes), _f(ys), B, classes.size(1), _f(self.output), _stream()),
                  "SelectBoxes")
        return self.output


class SoftMax(Module):
    def updateOutput(self, input):
        assert input.dim() == 2
        self.output = torch.empty_like(input)
        if input.size(0):
            check(_lib.load().mpn_softmax_forward(_f(input), input.size(0), input.size(1), _f(self.output), _stream()), "SoftMax")
        return self.output


class ImageTransformer(Module):
    """fbcoco.ImageTransformer(mean, std, scale, swap) — swap is 1-based like the Lua table."""

    def __init__(self, mean, std=None, scale=1, swap=None):
        super().__init__()
        self.mean, self.std, self.scale, self.swap = mean, std, scale or 1, swap

    def updateOutput(self, I):
        assert I.dim() == 3
        H, W = I.size(1), I.size(2)
        self.output = torch.empty((3, H, W), dtype=torch.float32, device=I.device)
        sw = (C.c_int * 3)(*[(s - 1) for s in (self.swap or (1, 2, 3))])
        mean = (C.c_double * 3)(*self.mean)
        std = (C.c_double * 3)(*self.std) if self.std else None
        check(_lib.load().mpn_image_transform(_f(I, "image"), H, W, sw, C.c_double(self.scale), mean, std, _f(self.output),
                                              _stream()), "ImageTransformer")
        return self.output


def RossTransformer():
    """model_utils.lua:138-140"""
    return ImageTransformer([102.9801, 115.9465, 122.7717], None, 255, [3, 2, 1])


def ImagenetTransformer():
    """model_utils.lua:143-155"""
    return ImageTransformer([0.48462227599918, 0.45624044862054, 0.40588363755159],
                            [0.22889466674951, 0.22446679341259, 0.22495548344775])


class SpatialConvolution(Module):
    """cudnn.SpatialConvolution(nIn, nOut, 3,3, 1,1, 1,1) with an optionally fused nn.ReLU."""

    def __init__(self, nInputPlane, nOutputPlane, kW=3, kH=3, dW=1, dH=1, padW=1, padH=1, relu=False):
        super().__init__()
        if (kW, kH, dW, dH, padW, padH) != (3, 3, 1, 1, 1, 1):
            raise NotImplementedError("the hot path only contains 3x3 / stride 1 / pad 1 convolutions (models/vgg.lua)")
        self.nInputPlane, self.nOutputPlane, self.relu = nInputPlane, nOutputPlane, relu
        self.weight = None  # [nOut, nIn, 3, 3]
        self.bias = None
        self._ws = None

    def updateOutput(self, input):
        x = input if input.dim() == 4 else input.unsqueeze(0)
        B, Cin, H, W = x.shape
        assert Cin == self.nInputPlane
        lib = _lib.load()
        need = lib.mpn_conv3x3_workspace_bytes(B, Cin, H, W, self.nOutputPlane)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        out = torch.empty((B, self.nOutputPlane, H, W), dtype=torch.float32, device=x.device)
        check(lib.mpn_conv3x3_forward(_f(x), B, Cin, H, W, _f(self.weight, "weight"),
                                      _f(self.bias, "bias") if self.bias is not None else None, self.nOutputPlane,
                                      int(self.relu), _f(out), C.c_void_p(self._ws.data_ptr()), C.c_size_t(need), _stream()),
              "SpatialConvolution")
        self.output = out if input.dim() == 4 else out[0]
        return self.output


class SpatialMaxPooling(Module):
    """nn.SpatialMaxPooling(2,2,2,2):ceil()"""

    def updateOutput(self, input):
        x = input.contiguous()
        H, W = x.shape[-2:]
        lead = x.shape[:-2]
        bc = 1
        for d in lead:
            bc *= d
        self.output = torch.empty(tuple(lead) + ((H + 1) // 2, (W + 1) // 2), dtype=torch.float32, device=x.device)
        check(_lib.load().mpn_maxpool2x2_ceil_forward(_f(x), bc, H, W, _f(self.output), _stream()), "SpatialMaxPooling")
        return self.output


class Linear(Module):
    """nn.Linear(inputSize, outputSize) with an optionally fused nn.ReLU."""

    def __init__(self, inputSize, outputSize, relu=False):
        super().__init__()
        self.inputSize, self.outputSize, self.relu = inputSize, outputSize, relu
        self.weight = None  # [out, in]
        self.bias = None

    def updateOutput(self, input):
        assert input.dim() == 2 and input.size(1) == self.inputSize
        M = input.size(0)
        self.output = torch.empty((M, self.outputSize), dtype=torch.float32, device=input.device)
        if M:
            check(_lib.load().mpn_linear_forward(_f(input), M, self.inputSize, _f(self.weight, "weight"),
                                                 _f(self.bias, "bias") if self.bias is not None else None, self.outputSize,
                                                 int(self.relu), _f(self.output), _stream()), "Linear")
        return self.output


class Sequential(Module):
    def __init__(self, *mods):
        super().__init__()
        self.modules = list(mods)

    def add(self, m):
        self.modules.append(m)
        return self

    def get(self, i):  # 1-based like Lua
        return self.modules[i - 1]

    def updateOutput(self, input):
        x = input
        for m in self.modules:
            x = m.updateOutput(x)
        self.output = x
        return x


class ModeSwitch(Module):
    """modules/ModeSwitch.lua:16-20 — evaluate mode runs the second branch."""

    def __init__(self, train_module, test_module):
        super().__init__()
        self.train = True
        self.modules = [train_module, test_module]

    def updateOutput(self, input):
        active = self.modules[0] if self.train else self.modules[1]
        self.output = active.updateOutput(input)
        return self.output
