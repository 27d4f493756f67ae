--[[ mpn.lua — LuaJIT FFI glue that drops libmpn_hip.so into the reference's Lua/Torch7 host code.

NOT RUN IN THIS REPOSITORY: the build image has no Lua / LuaJIT / Torch7 (DESIGN.md §2).  What IS checked, on CPU, by
tests/test_lua_binding.py: the cdef (mpn_cdef.lua) is regenerated from include/mpn.h and must match the committed file;
every prototype in it exists in libmpn_hip.so with the header's parameter list; every `C.mpn_*(...)` call below names a
declared function and passes the declared number of arguments; the struct fields set below exist.  The same entry points
are exercised through Python/ctypes on a real MI355X (tests/ -m gpu).

Three levels, least to most invasive (INTEGRATION.md):
  1. libnms.so            nothing to do here: utils.lua:15-26 loads it unchanged.
  2. module replacement   mpn.ROIPooling / mpn.Foveal / mpn.ContextRegion are nn.Modules with the reference's updateOutput
                          surface (inn.ROIPooling: models/vgg.lua:28; modules/Foveal.lua; modules/ContextRegion.lua):
                              model:replace(function(m)
                                 if torch.typename(m) == 'inn.ROIPooling' then return mpn.ROIPooling(m.W, m.H, m.spatial_scale) end
                                 if torch.typename(m) == 'nn.Foveal' then return mpn.Foveal() end
                                 return m end)
  3. whole path           mpn.FastRCNN(model, opt) builds the fused device pipeline from the nn.Sequential that models/vgg.lua
                          returns; its :testOne(im, boxes) is Tester_FRCNN:testOne (Tester_FRCNN.lua:54-139) and
                          mpn.Comm is the scored-box gather that replaces test_runner.lua:91-104.
]]
local ffi = require 'ffi'
ffi.cdef(require 'mpn_cdef')
ffi.cdef[[
void *THCState_getCurrentStream(void *state);                      /* libTHC: cutorch's current stream (a hipStream_t under ROCm) */
int hipMemcpy(void *dst, const void *src, size_t bytes, int kind); /* 2 = device to host */
]]
local C = ffi.load('libmpn_hip.so')
local mpn = {C = C}

local function check(rc, what)
   if rc ~= 0 then error(what .. ': ' .. ffi.string(C.mpn_last_error())) end
end

local function stream()  -- cutorch's current stream; nil = the default stream
   if cutorch and cutorch.getState then return ffi.C.THCState_getCurrentStream(cutorch.getState()) end
   return nil
end

-- inn.ROIPooling(W, H, spatial_scale) ------------------------------------------------------------------------------------
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

-- nn.Foveal (modules/Foveal.lua) — no D2H / Lua loop / H2D any more -------------------------------------------------------
local Foveal = torch.class('mpn.Foveal', 'nn.Module')
function Foveal:updateOutput(input)
   assert(input:nDimension() == 2)
   assert(input:size(2) == 5)
   self.output:resize(input:size(1) * 4, 5)
   check(C.mpn_foveal_forward(input:data(), input:size(1), self.output:data(), stream()), 'Foveal')
   return self.output
end

-- nn.ContextRegion(scale) (modules/ContextRegion.lua) ---------------------------------------------------------------------
local Context, cparent = torch.class('mpn.ContextRegion', 'nn.Module')
function Context:__init(scale) cparent.__init(self); self.scale = scale end
function Context:updateOutput(input)
   assert(input:nDimension() == 2)
   assert(input:size(2) == 5)
   self.output:resizeAs(input)
   check(C.mpn_context_region_forward(input:data(), input:size(1), self.scale, self.output:data(), stream()), 'ContextRegion')
   return self.output
end
mpn.ROIPooling, mpn.Foveal, mpn.ContextRegion = ROIPooling, Foveal, Context

-- Whole path: models/vgg.lua's graph as one device pipeline ------------------------------------------------------------------
-- model = nn.Sequential{ ParallelTable{features, Identity}, inn.ROIPooling, nn.View, classifier, ConcatTable{cls, bbox[+BBoxNorm]} }
-- opt: n_classes, max_h, max_w, max_rois, nms_thresh, num_iter, bbox_voting, ... (Tester_FRCNN.lua:20-51 fields)
local FastRCNN = torch.class('mpn.FastRCNN')
local function convs_of(features)  -- 3x3 convolutions in execution order + "a 2x2 max-pool follows" flags
   local convs, pool = {}, {}
   for _, m in ipairs(features:listModules()) do
      local tn = torch.typename(m)
      if tn == 'cudnn.SpatialConvolution' or tn == 'nn.SpatialConvolution' then
         assert(m.kW == 3 and m.kH == 3 and m.padW == 1 and m.dW == 1, 'the VGG pipeline takes 3x3 / stride 1 / pad 1 convolutions')
         convs[#convs + 1] = m; pool[#convs] = 0
      elseif tn == 'cudnn.SpatialMaxPooling' or tn == 'nn.SpatialMaxPooling' then pool[#convs] = 1 end
   end
   return convs, pool
end
function FastRCNN:__init(model, opt)
   local features, roipool = model:get(1):get(1), model:get(2)
   local top, heads = model:get(4), model:get(5)
   local lin = top:findModules('nn.Linear')                       -- fc6, fc7
   local cls, bbox = heads:get(1), heads:get(2)                   -- model_utils.lua:105-119
   local bnorm = bbox:findModules('nn.BBoxNorm')[1]
   cls, bbox = cls:findModules('nn.Linear')[1] or cls, bbox:findModules('nn.Linear')[1] or bbox
   local convs, pool = convs_of(features)
   local n = #convs
   local cfg = ffi.new('mpn_frcnn_config')
   self._cout, self._pool = ffi.new('int[?]', n), ffi.new('int[?]', n)
   local wp, bp = ffi.new('const float *[?]', n), ffi.new('const float *[?]', n)
   for i, m in ipairs(convs) do
      self._cout[i - 1], self._pool[i - 1] = m.nOutputPlane, pool[i]
      wp[i - 1], bp[i - 1] = m.weight:data(), m.bias:data()
   end
   cfg.n_conv, cfg.conv_cout, cfg.pool_after = n, self._cout, self._pool
   cfg.pooled_h, cfg.pooled_w, cfg.spatial_scale = roipool.H, roipool.W, roipool.spatial_scale
   cfg.fc_dim, cfg.n_classes = lin[2].weight:size(1), cls.weight:size(1)
   cfg.max_h, cfg.max_w, cfg.max_rois = opt.max_h or 600, opt.max_w or 1000, opt.max_rois or 2000
   cfg.tf_scale = 255                                             -- RossTransformer (model_utils.lua:138-140)
   local mean = {102.9801, 115.9465, 122.7717}
   for i = 0, 2 do cfg.tf_mean[i] = mean[i + 1]; cfg.tf_std[i] = 0; cfg.tf_swap[i] = 2 - i end
   if bnorm then for i = 0, 3 do cfg.bbox_mean[i] = bnorm.mean[i + 1]; cfg.bbox_std[i] = bnorm.std[i + 1] end end
   cfg.nms_thresh, cfg.score_thresh, cfg.top_k = opt.test_nms_threshold or 0.3, -1.5, 100
   cfg.num_iter, cfg.bbox_voting = opt.test_num_iterative_loc or 1, opt.test_bbox_voting and 1 or 0
   cfg.bbox_vote_thresh, cfg.bbox_vote_score_pow = opt.test_bbox_voting_nms_threshold or 0.5, opt.test_bbox_voting_score_pow or 1
   cfg.use_rbox_scores = opt.test_use_rbox_scores and 1 or 0
   cfg.scale_target, cfg.scale_max = opt.scale or 600, opt.max_size or 1000   -- getImages (ImageDetect.lua:34-43) on the device
   local h = ffi.new('mpn_frcnn *[1]')
   check(C.mpn_frcnn_create(cfg, wp, bp, lin[1].weight:data(), lin[1].bias:data(), lin[2].weight:data(), lin[2].bias:data(),
                            cls.weight:data(), cls.bias:data(), bbox.weight:data(), bbox.bias:data(), h), 'mpn_frcnn_create')
   self.handle = ffi.gc(h[0], C.mpn_frcnn_destroy)
   self.n_classes, self.top_cap = cfg.n_classes, 464
   self.dets, self.n_dets = torch.CudaTensor(self.top_cap, 6), torch.CudaIntTensor(1)
end
-- Tester:testOne (Tester_FRCNN.lua:54-139): im [3,H,W] and boxes [N,4] FloatTensors (host or device) -> img_boxes[j] = [K,5]
function FastRCNN:testOne(im, boxes)
   local d_im, d_boxes = im:cuda():contiguous(), boxes:cuda():contiguous()
   local s = stream()
   check(C.mpn_frcnn_test_one(self.handle, d_im:data(), im:size(2), im:size(3), d_boxes:data(), boxes:size(1), self.dets:data(),
                              self.top_cap, ffi.cast('int *', self.n_dets:data()), s), 'mpn_frcnn_test_one')
   local keep, idx, nk, stride = ffi.new('const float *[1]'), ffi.new('const int *[1]'), ffi.new('const int *[1]'), ffi.new('int[1]')
   check(C.mpn_frcnn_nms_results(self.handle, keep, idx, nk, stride), 'mpn_frcnn_nms_results')   -- synchronises
   local ncls, M = self.n_classes - 1, stride[0]
   local counts = ffi.new('int[?]', ncls)
   ffi.C.hipMemcpy(counts, nk[0], ncls * 4, 2)
   local img_boxes = {}
   for j = 1, ncls do
      local t = torch.FloatTensor(counts[j - 1], 5)
      if counts[j - 1] > 0 then ffi.C.hipMemcpy(t:data(), keep[0] + (j - 1) * M * 5, counts[j - 1] * 5 * 4, 2) end
      img_boxes[j] = t
   end
   return img_boxes
end
-- The latency mode (ModelParallelTable.lua:195-242's job for ONE image): every worker passes the SAME image and proposal table with
-- its own communicator (mpn.comm_init_all); each runs the trunk, the ROI head on its share of the proposals, the NMS of its share of
-- the classes, and two RCCL all-gathers of scored boxes connect the steps.  Every worker returns the same img_boxes — bit for bit what
-- testOne returns on one GPU.
function FastRCNN:testOneSharded(comm, im, boxes)
   local d_im, d_boxes = im:cuda():contiguous(), boxes:cuda():contiguous()
   check(C.mpn_frcnn_test_one_sharded(self.handle, comm, d_im:data(), im:size(2), im:size(3), d_boxes:data(), boxes:size(1), self.dets:data(),
                                      self.top_cap, ffi.cast('int *', self.n_dets:data()), stream()), 'mpn_frcnn_test_one_sharded')
   return self:_img_boxes()
end
function FastRCNN:_img_boxes()
   local keep, idx, nk, stride = ffi.new('const float *[1]'), ffi.new('const int *[1]'), ffi.new('const int *[1]'), ffi.new('int[1]')
   check(C.mpn_frcnn_nms_results(self.handle, keep, idx, nk, stride), 'mpn_frcnn_nms_results')   -- synchronises
   local ncls, M = self.n_classes - 1, stride[0]
   local counts = ffi.new('int[?]', ncls)
   ffi.C.hipMemcpy(counts, nk[0], ncls * 4, 2)
   local img_boxes = {}
   for j = 1, ncls do
      local t = torch.FloatTensor(counts[j - 1], 5)
      if counts[j - 1] > 0 then ffi.C.hipMemcpy(t:data(), keep[0] + (j - 1) * M * 5, counts[j - 1] * 5 * 4, 2) end
      img_boxes[j] = t
   end
   return img_boxes
end
-- utils.nms_dense (utils.lua:402-462): LongTensor of 1-based picks
function mpn.nms_dense(boxes, overlap)
   local n = boxes:nElement() == 0 and 0 or boxes:size(1)
   if n == 0 then return torch.LongTensor() end
   local d = boxes:cuda():contiguous()
   local pick, np_ = torch.CudaIntTensor(n), torch.CudaIntTensor(1)
   check(C.mpn_nms_dense(d:data(), n, overlap, ffi.cast('int *', pick:data()), ffi.cast('int *', np_:data()), stream()), 'mpn_nms_dense')
   local k = np_:int()[1]
   return k > 0 and pick:narrow(1, 1, k):long() or torch.LongTensor()
end
-- ImageDetect:detect (ImageDetect.lua:156-193): scores [N,C], decoded (unclamped) boxes [N,4C]; recompute_features as the reference
function FastRCNN:detect(im, boxes, recompute_features)
   local d_boxes = boxes:cuda():contiguous()
   local N = boxes:size(1)
   local scores, bbox = torch.CudaTensor(N, self.n_classes), torch.CudaTensor(N, 4 * self.n_classes)
   local d_im = (recompute_features == false) and nil or im:cuda():contiguous()
   check(C.mpn_frcnn_detect(self.handle, d_im and d_im:data() or nil, im:size(2), im:size(3), d_boxes:data(), N, scores:data(),
                            bbox:data(), 0, stream()), 'mpn_frcnn_detect')
   return scores, bbox
end
-- Captured launch graphs (off by default): the handle replays its kernel sequences with hipGraphLaunch while the caller keeps passing the
-- same device buffers and shapes.  Frees the host thread (one enqueue instead of ~100 per image); the GPU's timeline is unchanged.
function FastRCNN:setGraphs(on)
   check(C.mpn_frcnn_set_graphs(self.handle, on and 1 or 0), 'mpn_frcnn_set_graphs')
   return self
end
function FastRCNN:graphStats()
   local cap, rep = ffi.new('long[1]'), ffi.new('long[1]')
   check(C.mpn_frcnn_graph_stats(self.handle, cap, rep), 'mpn_frcnn_graph_stats')
   return tonumber(cap[0]), tonumber(rep[0])
end
mpn.FastRCNN = FastRCNN

-- Scored-box gather over RCCL (replaces test_runner.lua:91-104's per-thread result hand-back) -------------------------------------
-- one process, one worker thread per GPU (test_runner.lua:55-66): comms = mpn.comm_init_all(nGPU) in the main thread, worker i
-- (after cutorch.setDevice(i)) calls mpn.gather(comms[i - 1], net.dets, net.n_dets, out) once per image.
function mpn.comm_init_all(n)
   local comms = ffi.new('mpn_comm *[?]', n)
   check(C.mpn_comm_init_all(n, nil, comms), 'mpn_comm_init_all')
   return comms
end
-- what RCCL itself says the communicator spans (ncclCommCount, checked against ncclCommUserRank at init); 0 = no RCCL communicator
function mpn.comm_rccl_ranks(comm) return C.mpn_comm_rccl_ranks(comm) end
function mpn.gather(comm, dets, n_dets, out)  -- out: CudaTensor [world, top_cap*6 + 1]
   check(C.mpn_gather_dets(comm, dets:data(), ffi.cast('const int *', n_dets:data()), dets:size(1), out:data(), stream()), 'mpn_gather_dets')
   return out
end

return mpn
