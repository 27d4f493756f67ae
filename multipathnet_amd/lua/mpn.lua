--[[ mpn.lua — LuaJIT FFI glue that drops libmpn_hip.so into the reference's Lua/Torch7 host code.

UNTESTED IN THIS REPOSITORY: the build image has no Lua / LuaJIT / Torch7 (see DESIGN.md §2).  The file is
mechanically derived from include/mpn.h and mirrors what the Python host layer (multipathnet_amd/nn.py) does
through ctypes, which IS tested on the GPU.  Tensors are CudaTensors allocated by hipified cutorch (or any
allocator that yields device pointers); `:data()` gives the raw device pointer the C ABI wants.

Usage (replaces `require 'inn'` ROIPooling, modules/Foveal.lua, modules/ContextRegion.lua, and — through
libnms.so — utils.nms / utils.bbox_vote, with ImageDetect.lua and Tester_FRCNN.lua unchanged):

    local mpn = require 'mpn'
    model:replace(function(m)
       if torch.typename(m) == 'inn.ROIPooling' then return mpn.ROIPooling(m.W, m.H, m.spatial_scale) end
       if torch.typename(m) == 'nn.Foveal' then return mpn.Foveal() end
       return m
    end)
]]
local ffi = require 'ffi'

ffi.cdef[[
const char *mpn_last_error(void);
int mpn_roi_pool_forward(const float *d_feat, int B, int C, int H, int W, const float *d_rois, int N, int PH, int PW,
                         float scale, float coord_offset, int end_adjust, float *d_out, int32_t *d_argmax, void *stream);
int mpn_foveal_forward(const float *d_rois, int N, float *d_out, void *stream);
int mpn_context_region_forward(const float *d_rois, int N, double scale, float *d_out, void *stream);
int mpn_bbox_norm_forward(float *d_bbox, int N, int C4, const float *h_mean4, const float *h_std4, void *stream);
int mpn_select_boxes_forward(const float *d_scores, const float *d_bbox, int N, int C, float *d_out, void *stream);
int mpn_softmax_forward(const float *d_x, int M, int C, float *d_y, void *stream);
int mpn_bbox_decode(const float *d_boxes, const float *d_deltas, int N, int C, float *d_out, void *stream);
int mpn_nms_batched(const float *d_scored, const int *d_counts, int n_cls, int m_stride, float thr, float *d_keep,
                    int *d_keep_idx, int *d_n_keep, void *stream);
]]

local C = ffi.load('libmpn_hip.so')
local mpn = {}

local function check(rc, what)
   if rc ~= 0 then error(what .. ': ' .. ffi.string(C.mpn_last_error())) end
end

local function stream()  -- cutorch's current stream as a hipStream_t
   return cutorch and cutorch.getStream and ffi.cast('void*', cutorch._state_stream_ptr and cutorch._state_stream_ptr() or nil) or nil
end

-- inn.ROIPooling(W, H, spatial_scale) ------------------------------------------------------------
local ROIPooling, parent = torch.class('mpn.ROIPooling', 'nn.Module')
function ROIPooling:__init(W, H, spatial_scale)
   parent.__init(self)
   self.W, self.H, self.spatial_scale = W, H, spatial_scale or 1
   self.indices = torch.CudaIntTensor and torch.CudaIntTensor() or torch.CudaTensor()
end
function ROIPooling:setSpatialScale(s) self.spatial_scale = s; return self end
function ROIPooling:updateOutput(input)
   local feat, rois = input[1], input[2]
   assert(feat:nDimension() == 4 and rois:nDimension() == 2 and rois:size(2) == 5)
   local N = rois:size(1)
   self.output:resize(N, feat:size(2), self.H, self.W)
   self.indices:resize(N, feat:size(2), self.H, self.W)
   check(C.mpn_roi_pool_forward(feat:data(), feat:size(1), feat:size(2), feat:size(3), feat:size(4), rois:data(), N,
                                self.H, self.W, self.spatial_scale, 1.0, 0, self.output:data(),
                                ffi.cast('int32_t*', self.indices:data()), stream()), 'ROIPooling')
   return self.output
end

-- nn.Foveal (modules/Foveal.lua) — no D2H / Lua loop / H2D any more -----------------------------------
local Foveal, fparent = torch.class('mpn.Foveal', 'nn.Module')
function Foveal:updateOutput(input)
   assert(input:nDimension() == 2)
   assert(input:size(2) == 5)
   self.output:resize(input:size(1) * 4, 5)
   check(C.mpn_foveal_forward(input:data(), input:size(1), self.output:data(), stream()), 'Foveal')
   return self.output
end

-- nn.ContextRegion(scale) (modules/ContextRegion.lua) -----------------------------------------------
local Context, cparent = torch.class('mpn.ContextRegion', 'nn.Module')
function Context:__init(scale) cparent.__init(self); self.scale = scale end
function Context:updateOutput(input)
   assert(input:nDimension() == 2)
   assert(input:size(2) == 5)
   self.output:resizeAs(input)
   check(C.mpn_context_region_forward(input:data(), input:size(1), self.scale, self.output:data(), stream()), 'ContextRegion')
   return self.output
end

mpn.ROIPooling, mpn.Foveal, mpn.ContextRegion, mpn.C = ROIPooling, Foveal, Context, C
return mpn
