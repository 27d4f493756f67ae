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
                          returns; mpn.MultiPathNet(model, opt) from models/multipathnet.lua's (the system's namesake: skip pooling,
                          Foveal towers, integral classifiers), mpn.ResNet(model, opt) from models/resnet.lua's, mpn.Graph(model, opt)
                          from models/inceptionv3.lua's / models/alexnet.lua's.  All four are the same class: :testOne(im, boxes) is
                          Tester_FRCNN:testOne (Tester_FRCNN.lua:54-139), :detect is ImageDetect:detect, and mpn.gather is the
                          scored-box gather that replaces test_runner.lua:91-104.
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
-- bin_rule: 0 (default) = the module's CUDA branch, 1 = its CPU branch, crop + nn.SpatialAdaptiveMaxPooling (include/mpn.h MPN_ROI_BINS_*)
function ROIPooling:__init(W, H, spatial_scale, bin_rule)
   parent.__init(self)
   self.W, self.H, self.spatial_scale, self.bin_rule = W, H, spatial_scale or 1, bin_rule or 0
   self.indices = torch.CudaIntTensor and torch.CudaIntTensor() or torch.CudaTensor()
end
function ROIPooling:setSpatialScale(s) self.spatial_scale = s; return self end
function ROIPooling:updateOutput(input)
   local feat, rois = input[1], input[2]
   assert(feat:nDimension() == 4 and rois:nDimension() == 2 and rois:size(2) == 5)
   local N = rois:size(1)
   self.output:resize(N, feat:size(2), self.H, self.W)
   self.indices:resize(N, feat:size(2), self.H, self.W)
   check(C.mpn_roi_pool_forward_rule(feat:data(), feat:size(1), feat:size(2), feat:size(3), feat:size(4), rois:data(), N,
                                     self.H, self.W, self.spatial_scale, 1.0, 0, self.bin_rule or 0, self.output:data(),
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
local function is_conv(m) local tn = torch.typename(m); return tn == 'cudnn.SpatialConvolution' or tn == 'nn.SpatialConvolution' end
-- What wraps the reference's sub-networks: utils.makeDataParallel (model_utils.lua:15-29) returns nn.Sequential():add(module) at nGPU = 1 and
-- an nn.DataParallelTable of per-GPU clones otherwise; nn.NoBackprop (modules/NoBackprop.lua) is a one-child container;
-- utils.disableFeatureBackprop (model_utils.lua:96-103) moves the first layers into NoBackprop(nn.Sequential) at position 1.
-- replica(m): m without its outer wrappers, and ONE replica of it (the clones hold the same values) — so that findModules / listModules below
-- see every layer once, whatever nGPU was.  (Module:findModules returns the receiver itself first when it matches: never index its result to
-- "get inside" a wrapper.)
local function replica(m)
   while m.modules do
      local tn = torch.typename(m)
      if tn == 'nn.DataParallelTable' or tn == 'nn.NoBackprop' then m = m.modules[1]
      elseif tn == 'nn.Sequential' and #m.modules == 1 and m.modules[1].modules then m = m.modules[1]
      else break end
   end
   return m
end
-- conv_sequential(m): the nn.Sequential at or under m, in listModules' pre-order, whose FIRST module is a convolution — the container that holds
-- conv1 (ResNet's stem, inside disableFeatureBackprop's NoBackprop) / conv1_1 .. (VGG's skip_features)
local function conv_sequential(m)
   for _, s in ipairs(replica(m):listModules()) do
      if torch.typename(s) == 'nn.Sequential' and s.modules[1] and is_conv(s.modules[1]) then return s end
   end
   error('no nn.Sequential that starts with a convolution under ' .. torch.typename(m))
end
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
-- Tester_FRCNN.lua:20-51's option fields + the transformer -> the part of mpn_frcnn_config every model shares
local ROSS = {scale = 255, mean = {102.9801, 115.9465, 122.7717}, swap = {2, 1, 0}}                      -- model_utils.lua:138-140
local IMAGENET = {scale = 1, mean = {0.48462227599918, 0.45624044862054, 0.40588363755159},
                  std = {0.22889466674951, 0.22446679341259, 0.22495548344775}, swap = {0, 1, 2}}      -- model_utils.lua:143-155
local function common_config(cfg, opt, tf, bnorm)
   cfg.max_h, cfg.max_w, cfg.max_rois = opt.max_h or 600, opt.max_w or 1000, opt.max_rois or 2000
   cfg.tf_scale = tf.scale
   for i = 0, 2 do cfg.tf_mean[i] = tf.mean[i + 1]; cfg.tf_std[i] = tf.std and tf.std[i + 1] or 0; cfg.tf_swap[i] = tf.swap[i + 1] end
   if bnorm then for i = 0, 3 do cfg.bbox_mean[i] = bnorm.mean[i + 1]; cfg.bbox_std[i] = bnorm.std[i + 1] end end
   cfg.nms_thresh, cfg.score_thresh, cfg.top_k = opt.test_nms_threshold or 0.3, -1.5, opt.top_k or 100   -- Tester_FRCNN.lua:50,163
   cfg.num_iter, cfg.bbox_voting = opt.test_num_iterative_loc or 1, opt.test_bbox_voting and 1 or 0
   cfg.bbox_vote_thresh, cfg.bbox_vote_score_pow = opt.test_bbox_voting_nms_threshold or 0.5, opt.test_bbox_voting_score_pow or 1
   cfg.use_rbox_scores = opt.test_use_rbox_scores and 1 or 0
   cfg.roi_bin_rule = opt.roi_bin_rule or 0                        -- 1: inn.ROIPooling's CPU-branch bins (MPN_ROI_BINS_ADAPTIVE)
   cfg.fc_arith = opt.fc_arith or 0                                -- 1: fc6 as the three-plane bf16 split (MPN_FC_SPLIT3; auxiliary arithmetic)
   cfg.scale_target, cfg.scale_max = opt.scale or 600, opt.max_size or 1000   -- getImages (ImageDetect.lua:34-43) on the device
end
-- classAndBBoxLinear (model_utils.lua:105-119) [+ utils.integral's K classifier clones, model_utils.lua:275-317] -> (cls weight, cls bias,
-- bbox Linear, BBoxNorm or nil, K); the K clones are stacked into one [K*C, in] matrix, which is what the C ABI takes
local function heads_of(heads)
   local cls_part, bbox_part = heads:get(1), heads:get(2)
   local bnorm = bbox_part:findModules('nn.BBoxNorm')[1]
   local bbox = bbox_part:findModules('nn.Linear')[1] or bbox_part
   local cls = cls_part:findModules('nn.Linear')
   if #cls == 0 then cls = {cls_part} end
   local w, b = {}, {}
   for k, m in ipairs(cls) do w[k], b[k] = m.weight, m.bias end
   local cw, cb = (#cls == 1) and w[1] or torch.cat(w, 1), (#cls == 1) and b[1] or torch.cat(b, 1)
   return cw:contiguous(), cb:contiguous(), bbox, bnorm, #cls
end
-- the output buffers every model object holds: top_cap rows (keep_top_k keeps every row tied at the threshold: 4 k + 64, as the Python host)
local function finish(self, h, cfg, keep)
   self.handle = ffi.gc(h[0], C.mpn_frcnn_destroy)
   self._keep = keep                                              -- tensors / ffi arrays the create call read (freed with the object)
   self.n_classes, self.top_cap = cfg.n_classes, cfg.top_k * 4 + 64
   self.dets, self.n_dets = torch.CudaTensor(self.top_cap, 6), torch.CudaIntTensor(1)
end
function FastRCNN:__init(model, opt)
   local features, roipool = replica(model:get(1):get(1)), model:get(2)
   local lin = model:get(4):findModules('nn.Linear')             -- fc6, fc7
   local cls_w, cls_b, bbox, bnorm = heads_of(model:get(5))
   local convs, pool = convs_of(features)
   local n = #convs
   local cfg = ffi.new('mpn_frcnn_config')
   local cout, pl = ffi.new('int[?]', n), ffi.new('int[?]', n)
   local wp, bp = ffi.new('const float *[?]', n), ffi.new('const float *[?]', n)
   for i, m in ipairs(convs) do
      cout[i - 1], pl[i - 1] = m.nOutputPlane, pool[i]
      wp[i - 1], bp[i - 1] = m.weight:data(), m.bias:data()
   end
   cfg.n_conv, cfg.conv_cout, cfg.pool_after = n, cout, pl
   cfg.pooled_h, cfg.pooled_w, cfg.spatial_scale = roipool.H, roipool.W, roipool.spatial_scale
   cfg.fc_dim, cfg.n_classes = lin[2].weight:size(1), cls_w:size(1)
   common_config(cfg, opt, ROSS, bnorm)
   local h = ffi.new('mpn_frcnn *[1]')
   check(C.mpn_frcnn_create(cfg, wp, bp, lin[1].weight:data(), lin[1].bias:data(), lin[2].weight:data(), lin[2].bias:data(),
                            cls_w:data(), cls_b:data(), bbox.weight:data(), bbox.bias:data(), h), 'mpn_frcnn_create')
   finish(self, h, cfg, {cout, pl, cls_w, cls_b})
end

-- models/multipathnet.lua:30-120 — the namesake model.  model = nn.Sequential{
--   (1) ParallelTable{ NoBackprop(DataParallel(skip_features)), Identity }      skip_features: features 1-16, conv4 (17-23), conv5 (24-30) -> {conv5, conv4, conv3}
--   (2) ParallelTable{ Identity, Sequential{Foveal, View(-1,4,5), Transpose} }
--   (3) nn.ModelParallelTable(2) of region towers: Sequential{ ParallelTable{Identity, Select(1,i)}, conv345Combine, classifier:clone() } (+ the "het" tower)
--   (4) ConcatTable{Narrow, Narrow}   (5) classAndBBoxLinear [utils.integral: K classifier clones]   [(6) ModeSwitch] }
function mpn.MultiPathNet(model, opt)
   local self = setmetatable({}, {__index = FastRCNN})
   local skip = replica(model:get(1):get(1))                      -- NoBackprop(makeDataParallel(skip_features)) -> skip_features itself
   local conv_list, pool = convs_of(skip)                         -- listModules walks features 1-16, then conv4, then conv5: execution order
   local n = #conv_list
   local n3 = 0                                                    -- convolutions among skip_features' own first 16 layers = conv1_1 .. conv3_3
   local top = conv_sequential(skip)                               -- multipathnet.lua:34-57: 16 layers, ConcatTable{conv4, Identity}, ParallelTable, FlattenTable
   assert(#top.modules >= 17 and torch.typename(top:get(17)) == 'nn.ConcatTable', 'not models/multipathnet.lua\'s skip_features')
   for i = 1, 16 do local tn = torch.typename(top:get(i)); if tn == 'cudnn.SpatialConvolution' or tn == 'nn.SpatialConvolution' then n3 = n3 + 1 end end
   local n4 = n3
   for _, m in ipairs(top:get(17):get(1):listModules()) do local tn = torch.typename(m); if tn == 'cudnn.SpatialConvolution' or tn == 'nn.SpatialConvolution' then n4 = n4 + 1 end end
   local towers = model:get(3).modules
   local cls_w, cls_b, bbox, bnorm, K = heads_of(model:get(5))
   local cfg, mw = ffi.new('mpn_frcnn_config'), ffi.new('mpn_mpnet_weights')
   local cout, pl = ffi.new('int[?]', n), ffi.new('int[?]', n)
   local wp, bp = ffi.new('const float *[?]', n), ffi.new('const float *[?]', n)
   for i, m in ipairs(conv_list) do
      cout[i - 1], pl[i - 1] = m.nOutputPlane, pool[i]
      wp[i - 1], bp[i - 1] = m.weight:data(), m.bias:data()
   end
   pl[n - 1] = 0                                                   -- vgg.lua: no pool5
   cfg.n_conv, cfg.conv_cout, cfg.pool_after = n, cout, pl
   mw.n_towers, mw.tap_conv3, mw.tap_conv4, mw.n_integral = #towers, n3 - 1, n4 - 1, K
   local keep = {cout, pl, cls_w, cls_b}
   for t, tw in ipairs(towers) do
      local combine, classifier = tw:get(2), tw:get(3)
      mw.region[t - 1] = tw:get(1):get(2).index - 1               -- nn.Select(1, i): Foveal.lua:36-39's row i
      local use4, use3 = 0, 0
      for _, sel in ipairs(combine:findModules('nn.SelectTable')) do       -- conv345Combine's make1PoolingLayer(idx, ...): 2 = conv4, 3 = conv3
         if sel.index == 2 then use4 = 1 elseif sel.index == 3 then use3 = 1 end
      end
      mw.use_conv4[t - 1], mw.use_conv3[t - 1] = use4, use3
      if #combine:findModules('nn.Normalize') == 0 then mw.conv345_unnormalized = 1 end
      local mix = combine:findModules('cudnn.SpatialConvolution')[1] or combine:findModules('nn.SpatialConvolution')[1]
      local lin = classifier:findModules('nn.Linear')
      mw.mix_w[t - 1], mw.mix_b[t - 1] = mix.weight:data(), mix.bias:data()
      mw.fc6_w[t - 1], mw.fc6_b[t - 1] = lin[1].weight:data(), lin[1].bias:data()
      mw.fc7_w[t - 1], mw.fc7_b[t - 1] = lin[2].weight:data(), lin[2].bias:data()
      cfg.fc_dim = lin[2].weight:size(1)
      if t == 1 then local rp = combine:findModules('inn.ROIPooling')[1]; cfg.pooled_h, cfg.pooled_w, cfg.spatial_scale = rp.H, rp.W, rp.spatial_scale end
   end
   cfg.n_classes = cls_w:size(1) / K
   common_config(cfg, opt, ROSS, bnorm)
   local h = ffi.new('mpn_frcnn *[1]')
   check(C.mpn_mpnet_create(cfg, wp, bp, mw, cls_w:data(), cls_b:data(), bbox.weight:data(), bbox.bias:data(), h), 'mpn_mpnet_create')
   finish(self, h, cfg, keep)
   return self
end

-- BatchNorm already made a fixed affine by inn.utils.BNtoFixed / folded by inn.utils.foldBatchNorm (resnet.lua:34-36, inceptionv3.lua:23):
-- what is left of it behind a convolution (inn.ConstAffine: y = a x + b per channel) is folded into a COPY of the convolution's parameters
local function folded(conv, affine, keep)
   local w = conv.weight:clone():view(conv.nOutputPlane, -1)
   local b = conv.bias and conv.bias:clone() or w.new(conv.nOutputPlane):zero()
   if affine then
      w:cmul(affine.a:view(-1, 1):expandAs(w))
      b:cmul(affine.a):add(affine.b)
   end
   keep[#keep + 1], keep[#keep + 2] = w, b
   return w, b
end
local function is_affine(m) local tn = torch.typename(m); return tn == 'inn.ConstAffine' or tn == 'nn.SpatialBatchNormalization' or tn == 'cudnn.SpatialBatchNormalization' end

-- models/resnet.lua:24-50: net:get(1..7) on the image, inn.ROIPooling(14,14,1/16), net:get(8..10) = layer4 + average pool + View per ROI.
-- fb.resnet.torch blocks: Sequential{ ConcatTable{ Sequential{conv, bn, relu, ...conv, bn}, shortcut (Identity | Sequential{conv, bn}) }, CAddTable, ReLU }
function mpn.ResNet(model, opt)
   local self = setmetatable({}, {__index = FastRCNN})
   local features, roipool, classifier = replica(model:get(1):get(1)), model:get(2), replica(model:get(3))
   local cls_w, cls_b, bbox, bnorm = heads_of(model:get(4))
   local keep = {cls_w, cls_b}
   local W, B, cin, co, ks, st, pd, bn, bs = {}, {}, {}, {}, {}, {}, {}, {}, {}
   local function add_conv(seq_modules, i)                         -- convolution i of a Sequential + the affine that follows it, if any
      local m = seq_modules[i]
      local aff = seq_modules[i + 1] and is_affine(seq_modules[i + 1]) and seq_modules[i + 1] or nil
      assert(not aff or torch.typename(aff) == 'inn.ConstAffine', 'run inn.utils.BNtoFixed on the model first (resnet.lua:34-36)')
      local w, b = folded(m, aff, keep)
      local k = #W + 1
      W[k], B[k], cin[k], co[k], ks[k], st[k], pd[k] = w, b, m.nInputPlane, m.nOutputPlane, m.kW, m.dW, m.padW
   end
   local function add_blocks(container)                            -- every residual block under `container`, execution order
      for _, blk in ipairs(container:findModules('nn.ConcatTable')) do
         local path, short = blk:get(1), blk:get(2)
         local nc = 0
         for i, m in ipairs(path.modules) do if is_conv(m) then add_conv(path.modules, i); nc = nc + 1 end end
         local has = 0
         if torch.typename(short) ~= 'nn.Identity' then
            for i, m in ipairs(short.modules) do if is_conv(m) then add_conv(short.modules, i); has = 1 end end
         end
         bn[#bn + 1], bs[#bs + 1] = nc, has
      end
   end
   local stem = conv_sequential(features)                          -- resnet.lua:33: conv1 sits inside disableFeatureBackprop(features, 5)'s NoBackprop
   add_conv(stem.modules, 1)                                       -- conv1 7x7/2 (+ its BatchNorm unless inn.utils.foldBatchNorm folded it, resnet.lua:34)
   add_blocks(features)
   local n_trunk = #bn
   add_blocks(classifier)
   local n = #W
   local rw = ffi.new('mpn_resnet_weights')
   local wp, bp = ffi.new('const float *[?]', n), ffi.new('const float *[?]', n)
   local a_cin, a_co, a_ks, a_st, a_pd = ffi.new('int[?]', n), ffi.new('int[?]', n), ffi.new('int[?]', n), ffi.new('int[?]', n), ffi.new('int[?]', n)
   for k = 1, n do
      wp[k - 1], bp[k - 1] = W[k]:data(), B[k]:data()
      a_cin[k - 1], a_co[k - 1], a_ks[k - 1], a_st[k - 1], a_pd[k - 1] = cin[k], co[k], ks[k], st[k], pd[k]
   end
   local a_bn, a_bs = ffi.new('int[?]', #bn), ffi.new('int[?]', #bn)
   for k = 1, #bn do a_bn[k - 1], a_bs[k - 1] = bn[k], bs[k] end
   rw.n_convs, rw.w, rw.b, rw.cin, rw.cout, rw.ksize, rw.stride, rw.pad = n, wp, bp, a_cin, a_co, a_ks, a_st, a_pd
   rw.n_blocks, rw.block_n_convs, rw.block_has_shortcut, rw.n_trunk_blocks = #bn, a_bn, a_bs, n_trunk
   rw.n_heads, rw.n_integral, rw.bf16 = 1, 1, opt.bf16 and 1 or 0
   local cfg = ffi.new('mpn_frcnn_config')
   cfg.pooled_h, cfg.pooled_w, cfg.spatial_scale = roipool.H, roipool.W, roipool.spatial_scale
   cfg.n_classes, cfg.fc_dim = cls_w:size(1), cls_w:size(2)
   common_config(cfg, opt, IMAGENET, bnorm)
   local h = ffi.new('mpn_frcnn *[1]')
   check(C.mpn_resnet_create(cfg, rw, cls_w:data(), cls_b:data(), bbox.weight:data(), bbox.bias:data(), h), 'mpn_resnet_create')
   keep[#keep + 1] = {wp, bp, a_cin, a_co, a_ks, a_st, a_pd, a_bn, a_bs}
   finish(self, h, cfg, keep)
   return self
end

-- Branching graphs (models/inceptionv3.lua:27-43: net:get(1..25) | ROIPooling(17,17) @ 17/299 | net:get(26..30); models/alexnet.lua:14-27) as two
-- op lists (include/mpn.h mpn_graph_op).  The walker understands nn.Sequential, nn.Concat(2) / nn.DepthConcat (branches write side by side into
-- one tensor), convolutions (+ a following fixed affine, + a following ReLU), max / average pooling, SpatialCrossMapLRN, and the fc layers of
-- AlexNet's `top` (nn.Linear on the flattened pooled map = a convolution over the whole map); View / Dropout / Identity / Contiguous pass through.
local function graph_ops(root, c0, keep, pooled)
   local ops, tensor_c = {}, {c0}
   local function new_tensor(c) tensor_c[#tensor_c + 1] = c; return #tensor_c - 1 end
   local function walk(m, src, h)                                  -- emits m's ops reading tensor `src` (map side h, nil = unknown); returns its output tensor, map side
      local tn = torch.typename(m)
      if tn == 'nn.Sequential' then
         local i, mods = 1, m.modules
         while i <= #mods do
            local mm = mods[i]
            if is_conv(mm) or torch.typename(mm) == 'nn.Linear' then
               local aff = mods[i + 1] and is_affine(mods[i + 1]) and mods[i + 1] or nil
               local j = i + (aff and 1 or 0)
               local relu = mods[j + 1] and (torch.typename(mods[j + 1]) == 'nn.ReLU' or torch.typename(mods[j + 1]) == 'cudnn.ReLU')
               local op = {kind = 0, src = src, relu = relu and 1 or 0}
               if is_conv(mm) then
                  op.w, op.b = folded(mm, aff, keep)
                  op.cin, op.cout, op.kh, op.kw, op.sh, op.sw, op.ph, op.pw = mm.nInputPlane, mm.nOutputPlane, mm.kH, mm.kW, mm.dH, mm.dW, mm.padH, mm.padW
               else                                                -- nn.Linear(c * k * k, out) on a k x k map: the weight IS [out][c][k][k] (View(-1) order)
                  local k = h or 1
                  op.w, op.b = mm.weight, mm.bias
                  op.cin, op.cout, op.kh, op.kw, op.sh, op.sw, op.ph, op.pw = mm.weight:size(2) / (k * k), mm.weight:size(1), k, k, 1, 1, 0, 0
                  h = 1
               end
               op.dst = new_tensor(op.cout)
               ops[#ops + 1] = op
               src = op.dst
               i = j + (relu and 2 or 1)
            else
               src, h = walk(mm, src, h)
               i = i + 1
            end
         end
         return src, h
      elseif tn == 'nn.Concat' or tn == 'nn.DepthConcat' then
         local first, outs = #ops + 1, {}
         local total = 0
         for bi, br in ipairs(m.modules) do
            local o = walk(br, src, h)
            outs[bi] = {tensor = o, last = #ops}
            total = total + tensor_c[o + 1]
         end
         local dst, off = new_tensor(total), 0
         for _, o in ipairs(outs) do                                -- the branch's last op writes straight into the concatenated tensor
            local op = ops[o.last]
            assert(op.dst == o.tensor, 'a branch of a Concat must end in a convolution or a pooling layer')
            op.dst, op.dst_c_off = dst, off
            off = off + tensor_c[o.tensor + 1]
         end
         return dst, h
      elseif tn == 'nn.SpatialMaxPooling' or tn == 'cudnn.SpatialMaxPooling' or tn == 'nn.SpatialAveragePooling' or tn == 'cudnn.SpatialAveragePooling' then
         local c = tensor_c[src + 1]
         local op = {kind = tn:find('Max') and 1 or 2, src = src, cin = c, cout = c, kh = m.kH, kw = m.kW, sh = m.dH, sw = m.dW, ph = m.padH or 0, pw = m.padW or 0,
                     ceil_mode = m.ceil_mode and 1 or 0, relu = 0}
         op.dst = new_tensor(c)
         ops[#ops + 1] = op
         return op.dst, h
      elseif tn == 'nn.SpatialCrossMapLRN' or tn == 'inn.SpatialCrossResponseNormalization' then
         local c = tensor_c[src + 1]
         local op = {kind = 3, src = src, cin = c, cout = c, kh = m.size, kw = 1, sh = 1, sw = 1, ph = 0, pw = 0, relu = 0,
                     lrn_alpha = m.alpha, lrn_beta = m.beta, lrn_k = m.k or 1}
         op.dst = new_tensor(c)
         ops[#ops + 1] = op
         return op.dst, h
      elseif tn == 'nn.DataParallelTable' or (m.modules and #m.modules == 1) then   -- NoBackprop / makeDataParallel wrappers: one replica
         return walk(m.modules[1], src, h)
      end
      return src, h                                                 -- View, Dropout, Identity, Contiguous, a ReLU that follows a pooling layer
   end
   local out = walk(root, 0, pooled)
   local arr, tc = ffi.new('mpn_graph_op[?]', #ops), ffi.new('int[?]', #tensor_c)
   for i, c in ipairs(tensor_c) do tc[i - 1] = c end
   for i, o in ipairs(ops) do
      local op = arr[i - 1]
      op.kind, op.src, op.dst, op.dst_c_off, op.src_c_off = o.kind, o.src, o.dst, o.dst_c_off or 0, 0
      op.cin, op.cout, op.kh, op.kw, op.sh, op.sw, op.ph, op.pw, op.relu = o.cin, o.cout, o.kh, o.kw, o.sh, o.sw, o.ph, o.pw, o.relu
      op.ceil_mode, op.lrn_alpha, op.lrn_beta, op.lrn_k = o.ceil_mode or 0, o.lrn_alpha or 0, o.lrn_beta or 0, o.lrn_k or 0
      if o.w then op.w, op.b = o.w:data(), o.b:data() end
   end
   keep[#keep + 1] = {arr, tc}
   return arr, #ops, tc, #tensor_c, out
end
-- opt.transformer: {scale, mean, std, swap} (default: fbcoco.ImageTransformer({1,1,1}, nil, 2), inceptionv3.lua:52; pass mpn.ROSS for alexnet.lua)
function mpn.Graph(model, opt)
   local self = setmetatable({}, {__index = FastRCNN})
   local features, roipool, classifier = replica(model:get(1):get(1)), model:get(2), replica(model:get(3))
   -- (grouped convolutions — alexnet.lua's conv2 / conv4 / conv5 — must be split into one convolution per group by the caller: op.src_c_off)
   local cls_w, cls_b, bbox, bnorm = heads_of(model:get(#model.modules))
   local keep = {cls_w, cls_b}
   local gw = ffi.new('mpn_graph_weights')
   local t_ops, t_n, t_tc, t_nt, feat = graph_ops(features, 3, keep)
   local feat_c = t_tc[feat]
   local h_ops, h_n, h_tc, h_nt, out = graph_ops(classifier, feat_c, keep, roipool.H)
   gw.n_trunk_ops, gw.trunk_ops, gw.n_trunk_tensors, gw.trunk_tensor_c, gw.feat_tensor = t_n, t_ops, t_nt, t_tc, feat
   gw.n_head_ops, gw.head_ops, gw.n_head_tensors, gw.head_tensor_c, gw.out_tensor = h_n, h_ops, h_nt, h_tc, out
   gw.bf16, gw.n_heads, gw.n_integral = opt.bf16 and 1 or 0, 1, 1
   local cfg = ffi.new('mpn_frcnn_config')
   cfg.pooled_h, cfg.pooled_w, cfg.spatial_scale = roipool.H, roipool.W, roipool.spatial_scale
   cfg.n_classes, cfg.fc_dim = cls_w:size(1), cls_w:size(2)
   common_config(cfg, opt, opt.transformer or {scale = 2, mean = {1, 1, 1}, swap = {0, 1, 2}}, bnorm)
   local h = ffi.new('mpn_frcnn *[1]')
   check(C.mpn_graph_create(cfg, gw, cls_w:data(), cls_b:data(), bbox.weight:data(), bbox.bias:data(), h), 'mpn_graph_create')
   finish(self, h, cfg, keep)
   return self
end
mpn.ROSS, mpn.IMAGENET = ROSS, IMAGENET

-- Tester:testOne (Tester_FRCNN.lua:54-139): im [3,H,W] and boxes [N,4] FloatTensors (host or device) -> img_boxes[j] = [K,5]
function FastRCNN:testOne(im, boxes)
   local d_im, d_boxes = im:cuda():contiguous(), boxes:cuda():contiguous()
   check(C.mpn_frcnn_test_one(self.handle, d_im:data(), im:size(2), im:size(3), d_boxes:data(), boxes:size(1), self.dets:data(),
                              self.top_cap, ffi.cast('int *', self.n_dets:data()), stream()), 'mpn_frcnn_test_one')
   return self:_img_boxes()
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
