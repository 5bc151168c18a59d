--[[ catgan.nn — Torch7's nn / cudnn / stn classes of the hot path re-created over include/catgan.h.

Same class names, constructor signatures and nn.Module protocol the reference relies on (SURVEY.md 8b; the in-tree
LeakyReLU.lua:5-31 and layers/SpatialConvolutionUpsample.lua:1-56 show the upstream shape of it): __init,
updateOutput / updateGradInput / accGradParameters, forward / backward, .weight .bias .gradWeight .gradBias .output
.gradInput .modules .__typename, :add :get :size :listModules :parameters :getParameters :zeroGradParameters
:training :evaluate :clone :cuda :float :type.  With
    torch = require 'catgan'.torch ; nn = require 'catgan'.nn ; cudnn = require 'catgan'.cudnn
in place of the Torch7 rocks, models.lua:138-160 / :196-228 / :640-711 / :814-906 build as written.

Two ways through a network, as in the executable twin cat-generator_amd/nn.py:
  * the nn.Module protocol module by module - one C-ABI call (or a short fixed sequence) per module method: what a module used
    on its own gets;
  * planned passes: nn.Sequential:forward / :backward / :updateGradInput on a whole network (what adversarial.lua:84-89,182-197
    calls on MODEL_D / MODEL_G) describe the tree to the library once and run as ONE cg_net_forward / cg_net_backward call
    (catgan.net; csrc/net.hip plans fusion, lockstep branches, streams, re-packing below the C ABI).  nn.planned = false keeps
    the per-module walk.
tools/abi_replay replays both call sequences without any interpreter in the process (tests/test_abi_step.py).  Feature maps stay
NHWC between modules; nn.View / nn.Transpose / nn.Copy are where the logical Torch7 layout is (re)established.
]]
local ffi = require 'ffi'
local abi = require 'catgan.ffi'
local T = require 'catgan.tensor'
local C, check = abi.C, abi.check
local Device, Host, prod, copy_shape = T.Device, T.Host, T.prod, T.copy_shape

local nn, cudnn = {}, {}
nn.planned = os.getenv('CG_PLANNED') ~= '0'
local function S() return T.stream end

-- counter-based generator shared with the engine's kernels (cg_rng_*): seed + running offset
local rng = { seed = 1, offset = 0 }
function rng.take(n) local o = rng.offset; rng.offset = o + n; return o end
nn.rng = rng

-- one grow-only device scratch (split-K partials, slope-gradient partials)
local scratch = { ptr = nil, bytes = 0 }
local function workspace(bytes)
   bytes = math.max(tonumber(bytes), 4096)
   if scratch.bytes < bytes then
      check(C.cg_stream_sync(S()))   -- nobody may still be using the old block
      local p = ffi.new('void*[1]'); check(C.cg_malloc(p, bytes * 5 / 4))
      scratch.ptr, scratch.bytes = ffi.gc(p[0], function(q) C.cg_free(q) end), bytes * 5 / 4
   end
   return scratch.ptr, scratch.bytes
end

local function to_device(x)   -- accept a host FloatTensor where Torch7 would accept one
   if x.__typename == 'torch.FloatTensor' then return Device.new(x.shape):copy(x) end
   return x
end

-- ------------------------------------------------------------------------------------------------ class helper
local Module = { __typename = 'nn.Module' }
Module.__index = Module
local function class(name, parent)
   parent = parent or Module
   local c = setmetatable({ __typename = name, __parent = parent }, {
      __index = parent,
      __call = function(cls, ...) local o = setmetatable({}, cls); o:__init(...); return o end })
   c.__index = c
   c.__tostring = function(self) return self:__repr() end
   return c
end
nn.Module, nn.class = Module, class

function Module:__init() self.output = nil; self.gradInput = nil; self.train = true; self._bufs = {} end
function Module:__repr() return self.__typename end
-- one persistent buffer per (role, element count): the half-batch (fake generation) and full-batch passes of a module
-- keep separate storage
function Module:_buf(role, shape, fmt)
   local n = prod(shape)
   local key = role .. ':' .. n
   local b = self._bufs[key]
   if not b then
      b = (fmt == 'nhwc') and Device.new_nhwc(shape) or Device.new(shape)
      self._bufs[key] = b
   else
      b.shape, b.fmt, b.ups = copy_shape(shape), fmt or 'plain', 0
   end
   return b
end
function Module:updateOutput(input) error(self.__typename .. ':updateOutput not implemented') end
function Module:updateGradInput(input, gradOutput) error(self.__typename .. ':updateGradInput not implemented') end
function Module:accGradParameters(input, gradOutput, scale) end
function Module:forward(input) return self:updateOutput(input) end
function Module:backward(input, gradOutput, scale)
   scale = scale or 1
   self:updateGradInput(input, gradOutput)
   self:accGradParameters(input, gradOutput, scale)
   return self.gradInput
end
function Module:listModules() return { self } end
function Module:training() for _, m in ipairs(self:listModules()) do m.train = true end; return self end
function Module:evaluate() for _, m in ipairs(self:listModules()) do m.train = false end; return self end
-- :cuda() moves host-side parameter tensors to the device (models.lua:860 REPLACES the classifier's bias by a
-- torch.Tensor(init_bias), a host FloatTensor, before conv:cuda() at :706); everything else already lives there
function Module:cuda()
   for _, m in ipairs(self:listModules()) do
      for _, ref in ipairs(m:own_parameters()) do
         local w = m[ref[2]]
         if w.__typename == 'torch.FloatTensor' then
            m[ref[2]] = Device.new(w.shape):copy(w)
            m[ref[3]] = Device.new(w.shape):zero()
         end
      end
   end
   return self
end
function Module:float() return self end
function Module:type() return self end
function Module:reset() end
-- node:clearState() (utils/nn_utils.lua:429, NN_UTILS.prepareNetworkForSave before torch.save): drop everything a pass left behind -
-- outputs, gradInputs, masks, persistent buffers, packed weight copies, the compiled plan - so that a checkpoint holds parameters
-- and constructor fields only.  The next forward rebuilds them.
local TRANSIENT = { 'output', 'gradInput', 'noise', 'finput', 'fgradInput', '_planned_last', '_pnet', '_pnet_n', '_wf', '_wb', '_wb_ph', '_u_bwd',
                    '_ph_version', '_ph_ptr', '_packed_ptr', '_loss', '_in_shape', '_g', '_bsums_g', '_sizes', '_count', 'save_std', 'save_mean' }
function Module:clearState()
   for _, m in ipairs(self:listModules()) do
      for _, k in ipairs(TRANSIENT) do rawset(m, k, nil) end
      m._bufs = {}
   end
   return self
end
function Module:own_parameters()   -- { {module, 'weight', 'gradWeight'}, ... } in Torch7 order (weight, then bias)
   local out = {}
   if self.weight then out[#out + 1] = { self, 'weight', 'gradWeight' } end
   if self.bias then out[#out + 1] = { self, 'bias', 'gradBias' } end
   return out
end
function Module:parameters()
   local w, g = {}, {}
   for _, m in ipairs(self:listModules()) do
      for _, ref in ipairs(m:own_parameters()) do w[#w + 1] = ref[1][ref[2]]; g[#g + 1] = ref[1][ref[3]] end
   end
   return w, g
end
-- Module:getParameters() (train.lua:184-185): one contiguous vector per net, depth-first, weight then bias; every
-- weight / bias / gradWeight / gradBias becomes a view into it (this is what makes ONE all-reduce per net possible)
function Module:getParameters()
   local refs, n = {}, 0
   for _, m in ipairs(self:listModules()) do
      for _, ref in ipairs(m:own_parameters()) do refs[#refs + 1] = ref; n = n + ref[1][ref[2]].n end
   end
   local flat, gflat = Device.new(n):zero(), Device.new(n):zero()
   local off = 0
   for _, ref in ipairs(refs) do
      local m, p, g = ref[1], ref[2], ref[3]
      local w = m[p]
      local view = Device.view_of(flat.store, off, w.shape, 'plain', 0)
      if w.__typename == 'torch.FloatTensor' then view:copy(w)   -- a host tensor someone assigned (models.lua:860)
      else check(C.cg_memcpy_d2d(S(), view.ptr, w.ptr, w.n * 4)) end
      m[p] = view
      m[g] = Device.view_of(gflat.store, off, w.shape, 'plain', 0)
      off = off + w.n
   end
   flat:touch()
   return flat, gflat
end
function Module:zeroGradParameters()
   local _, g = self:parameters()
   for _, t in ipairs(g) do t:zero() end
end
-- net:clone() (utils/nn_utils.lua:629): deep copy of the module tree, parameters included
local function deep(o, seen)
   if type(o) ~= 'table' then return o end
   if seen[o] then return seen[o] end
   if o.__tensor then local c = o:clone(); seen[o] = c; return c end
   local c = {}; seen[o] = c
   for k, v in pairs(o) do if k ~= '_bufs' then c[k] = deep(v, seen) end end
   if o._bufs then c._bufs = {} end
   return setmetatable(c, getmetatable(o))
end
function Module:clone() return deep(self, {}) end

-- ------------------------------------------------------------------------------------------------- containers
local Sequential = class('nn.Sequential')
function Sequential:__init() Module.__init(self); self.modules = {} end
function Sequential:add(m) self.modules[#self.modules + 1] = m; return self end
function Sequential:get(i) return self.modules[i] end
function Sequential:size() return #self.modules end
function Sequential:listModules()
   local out = { self }
   for _, m in ipairs(self.modules) do for _, c in ipairs(m:listModules()) do out[#out + 1] = c end end
   return out
end
-- planned passes (catgan.net): the calls adversarial.lua makes on a whole network
local function planned_net(self)
   if not nn.planned or self.__typename ~= 'nn.Sequential' then return nil end
   if self._pnet == nil or self._pnet_n ~= #self:listModules() then
      self._pnet = require('catgan.net').Net.new(self) or false
      self._pnet_n = #self:listModules()
   end
   return self._pnet or nil
end
function Sequential:forward(input)
   local net = planned_net(self)
   if not net then return self:updateOutput(input) end
   local x = to_device(input):materialise()
   local out = net:forward(x, rng, nil)
   self._planned_last = true
   local last = self.modules[#self.modules]
   if last and last.__typename == 'nn.Copy' and not last.outtype:find('Cuda') then out = out:float() end   -- models.lua:704
   self.output = out
   return out
end
-- Both generator forwards of an iteration side by side (a six-line change to adversarial.lua, INTEGRATION.md section 1): planned nets only
function Sequential:forwardPair(input, input2)
   local net = assert(planned_net(self), 'forwardPair needs the planned executor')
   local out = net:forward_pair(to_device(input):materialise(), to_device(input2):materialise(), rng, nil)
   self._planned_last = true
   return out
end
function Sequential:pairJoin()
   self.output = self._pnet:pair_join()
   return self.output
end
-- data parallelism: join the gradient buckets the planned backward started (self._bucket_overlap = true before the first pass); 0 = none
-- travelled (per-module walk, or parameters that are not contiguous) and the host all-reduces the whole vector itself
function Sequential:finishBuckets(comm)
   if not (self._pnet and self._planned_last) then return 0 end
   return self._pnet:finish_buckets(comm)
end
local function planned_backward(self, gradOutput, scale, acc)
   local gi = self._pnet:backward(to_device(gradOutput):materialise(), acc, scale)
   local first = self.modules[1]
   if first then first.gradInput = gi end                     -- adversarial.lua:193 reads MODEL_D.modules[1].gradInput
   if first and first.__typename == 'nn.Copy' and not first.intype:find('Cuda') then gi = gi:float() end
   self.gradInput = gi
   return gi
end
function Sequential:updateOutput(input)
   self._planned_last = false
   local cur = input
   for _, m in ipairs(self.modules) do cur = m:updateOutput(cur) end
   self.output = cur
   return cur
end
local function walk_back(self, input, gradOutput, fn)
   local cur = gradOutput
   for i = #self.modules, 2, -1 do cur = fn(self.modules[i], self.modules[i - 1].output, cur) end
   cur = fn(self.modules[1], input, cur)
   self.gradInput = cur
   return cur
end
function Sequential:updateGradInput(input, gradOutput)
   if self._planned_last then return planned_backward(self, gradOutput, 1, false) end
   return walk_back(self, input, gradOutput, function(m, i, g) return m:updateGradInput(i, g) end)
end
function Sequential:accGradParameters(input, gradOutput, scale)
   local cur = gradOutput
   for i = #self.modules, 2, -1 do
      self.modules[i]:accGradParameters(self.modules[i - 1].output, cur, scale)
      cur = self.modules[i].gradInput
   end
   self.modules[1]:accGradParameters(input, cur, scale)
end
function Sequential:backward(input, gradOutput, scale)
   scale = scale or 1
   if self._planned_last then return planned_backward(self, gradOutput, scale, true) end
   return walk_back(self, input, gradOutput, function(m, i, g) return m:backward(i, g, scale) end)
end
function Sequential:__repr()
   local s = { 'nn.Sequential {' }
   for i, m in ipairs(self.modules) do s[#s + 1] = '  (' .. i .. '): ' .. tostring(m):gsub('\n', '\n  ') end
   s[#s + 1] = '}'
   return table.concat(s, '\n')
end
nn.Sequential = Sequential

-- nn.ConcatTable(): every branch sees the input; output = table of branch outputs; backward sums the gradInputs
local ConcatTable = class('nn.ConcatTable', Sequential)
function ConcatTable:updateOutput(input)
   self._planned_last = false
   self.output = {}
   for i, m in ipairs(self.modules) do self.output[i] = m:updateOutput(input) end
   return self.output
end
local function sum_grads(self, grads)
   local acc
   for _, g in ipairs(grads) do
      if acc == nil then acc = g
      else
         local a, b = (#acc.shape == 4) and acc:nhwc() or acc, (#g.shape == 4) and g:nhwc() or g
         local out = self:_buf('sum', a.shape, a.fmt)
         check(C.cg_add(S(), a.ptr, b.ptr, out.ptr, a:phys_n()))
         acc = out
      end
   end
   self.gradInput = acc
   return acc
end
function ConcatTable:updateGradInput(input, gradOutput)
   local gs = {}
   for i, m in ipairs(self.modules) do gs[i] = m:updateGradInput(input, gradOutput[i]) end
   return sum_grads(self, gs)
end
function ConcatTable:accGradParameters(input, gradOutput, scale)
   for i, m in ipairs(self.modules) do m:accGradParameters(input, gradOutput[i], scale) end
end
function ConcatTable:backward(input, gradOutput, scale)
   local gs = {}
   for i, m in ipairs(self.modules) do gs[i] = m:backward(input, gradOutput[i], scale or 1) end
   return sum_grads(self, gs)
end
nn.ConcatTable = ConcatTable

-- nn.Concat(2) (models.lua:688-692): branch outputs joined on channels
local Concat = class('nn.Concat', Sequential)
function Concat:__init(dimension)
   Sequential.__init(self); assert(dimension == 2, 'only channel concatenation is on the path'); self.dimension = dimension
end
function Concat:updateOutput(input)
   self._planned_last = false
   local outs, Ct = {}, 0
   self._sizes = {}
   for i, m in ipairs(self.modules) do
      outs[i] = m:updateOutput(input):nhwc()
      self._sizes[i] = outs[i].shape[2]; Ct = Ct + self._sizes[i]
   end
   local N, H, W = outs[1].shape[1], outs[1].shape[3], outs[1].shape[4]
   local out = self:_buf('out', { N, Ct, H, W }, 'nhwc')
   local off = 0
   for i, o in ipairs(outs) do
      check(C.cg_copy_channels(S(), o.ptr, out.ptr, N * H * W, self._sizes[i], 0, Ct, off, self._sizes[i]))
      off = off + self._sizes[i]
   end
   self.output = out
   return out
end
local function concat_back(self, input, gradOutput, fn)
   local g = gradOutput:nhwc()
   local N, Ct, H, W = g.shape[1], g.shape[2], g.shape[3], g.shape[4]
   local off, acc = 0, nil
   for i, m in ipairs(self.modules) do
      local c = self._sizes[i]
      local s = self:_buf('gslice' .. i, { N, c, H, W }, 'nhwc')
      check(C.cg_copy_channels(S(), g.ptr, s.ptr, N * H * W, Ct, off, c, 0, c))
      off = off + c
      local gi = fn(m, s):nhwc()
      if acc == nil then
         acc = self:_buf('gsum', gi.shape, 'nhwc'); acc:copy(gi)
      else
         check(C.cg_axpy(S(), 1.0, gi.ptr, acc.ptr, acc:phys_n()))
      end
   end
   self.gradInput = acc
   return acc
end
function Concat:updateGradInput(input, gradOutput)
   return concat_back(self, input, gradOutput, function(m, s) return m:updateGradInput(input, s) end)
end
function Concat:backward(input, gradOutput, scale)
   return concat_back(self, input, gradOutput, function(m, s) return m:backward(input, s, scale or 1) end)
end
nn.Concat = Concat

-- ----------------------------------------------------------------------------------- convolution / linear (GEMM)
-- canonical parameters (Torch7 layout) + packed copies for the kernels, refreshed when the flat vector changed
local Gemm = class('nn._GemmLayer')
function Gemm:_ensure_packed()
   local v = self.weight.store.version
   if self._packed_version == v and self._packed_ptr == self.weight.ptr then return end
   local Cout, Cin, kH, kW = self:_wdims()
   local n = Cout * Cin * kH * kW
   self._wf = self._wf or Device.new(n)
   if kH * kW > 1 then self._wb = self._wb or Device.new(n) end
   check(C.cg_pack_conv_weight(S(), self.weight.ptr, self._wf.ptr, self._wb and self._wb.ptr or nil, Cout, Cin, kH, kW))
   self._packed_version, self._packed_ptr = v, self.weight.ptr
end
function Gemm:_ensure_packed_ups()   -- phase-summed weights for upsample2 -> conv (and their Winograd-domain form)
   local v = self.weight.store.version
   if self._ph_version == v and self._ph_ptr == self.weight.ptr then return end
   local Cout, Cin, k = self.nOutputPlane, self.nInputPlane, self.kH
   local pad = (k - 1) / 2
   local n = tonumber(C.cg_pack_conv_weight_ups2_floats(Cout, Cin, k, pad))
   self._wf_ph = self._wf_ph or Device.new(n); self._wb_ph = self._wb_ph or Device.new(n)
   check(C.cg_pack_conv_weight_ups2(S(), self.weight.ptr, self._wf_ph.ptr, self._wb_ph.ptr, Cout, Cin, k, pad))
   if self._wino then
      local nu = tonumber(C.cg_conv2d_ups2_wino_u_floats(Cin, Cout))
      self._u_fwd = self._u_fwd or Device.new(nu); self._u_bwd = self._u_bwd or Device.new(nu)
      check(C.cg_conv2d_ups2_wino_pack(S(), self._wf_ph.ptr, self._wb_ph.ptr, self._u_fwd.ptr, self._u_bwd.ptr, Cout, Cin))
   end
   self._ph_version, self._ph_ptr = v, self.weight.ptr
end
-- nn.Linear:reset / nn.SpatialConvolution:reset [upstream]: U(+-stdv*sqrt(3)) if stdv is given, else U(+-1/sqrt(fan_in)),
-- for weight and bias; drawn from the engine's counter stream (bit-equal to the Python host and the oracle)
function Gemm:reset(stdv)
   local s = stdv and stdv * math.sqrt(3) or 1 / math.sqrt(self:_fan_in())
   check(C.cg_rng_uniform(S(), self.weight.ptr, self.weight.n, -s, s, rng.seed, rng.take(self.weight.n))); self.weight:touch()
   check(C.cg_rng_uniform(S(), self.bias.ptr, self.bias.n, -s, s, rng.seed, rng.take(self.bias.n))); self.bias:touch()
   return self
end
local function conv_forward(x, wf, bias, out, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups)
   local ws, wsb = workspace(C.cg_conv2d_workspace_bytes(N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups))
   check(C.cg_conv2d_forward(S(), x.ptr, wf, bias, out.ptr, N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups, ws, wsb))
end

-- nn.Linear(inputSize, outputSize)  (models.lua:199,697,700,850,853)
local Linear = class('nn.Linear', Gemm)
function Linear:__init(i, o)
   Module.__init(self)
   self.weight = Device.new(o, i); self.bias = Device.new(o)
   self.gradWeight = Device.new(o, i):zero(); self.gradBias = Device.new(o):zero()
   self:reset()
end
function Linear:_wdims() return self.weight.shape[1], self.weight.shape[2], 1, 1 end
function Linear:_fan_in() return self.weight.shape[2] end
function Linear:updateOutput(input)
   local x = to_device(input):plain()
   local N, i, o = x.shape[1], x.shape[2], self.weight.shape[1]
   self:_ensure_packed()
   local out = self:_buf('out', { N, o })
   conv_forward(x, self._wf.ptr, self.bias.ptr, out, N, 1, 1, i, o, 1, 1, 0, 0, 0)
   self._x, self.output = x, out
   return out
end
function Linear:updateGradInput(input, gradOutput)
   local dy = gradOutput:plain()
   local N, o, i = dy.shape[1], dy.shape[2], self.weight.shape[2]
   local gi = self:_buf('gin', { N, i })
   conv_forward(dy, self.weight.ptr, nil, gi, N, 1, 1, o, i, 1, 1, 0, 0, 0)   -- canonical [out][in] IS the backward operand
   self.gradInput = gi
   return gi
end
function Linear:accGradParameters(input, gradOutput, scale)
   local x, dy = self._x, gradOutput:plain()
   local N, i, o = x.shape[1], x.shape[2], self.weight.shape[1]
   local ws, wsb = workspace(C.cg_conv2d_wgrad_workspace_bytes(N, 1, 1, i, o, 1, 1, 0, 0, 0))
   check(C.cg_conv2d_wgrad(S(), x.ptr, dy.ptr, self.gradWeight.ptr, self.gradBias.ptr, N, 1, 1, i, o, 1, 1, 0, 0, 0, scale or 1, ws, wsb))
end
function Linear:__repr() return ('nn.Linear(%d -> %d)'):format(self.weight.shape[2], self.weight.shape[1]) end
nn.Linear = Linear

-- nn.SpatialConvolution(nIn, nOut, kW, kH, dW, dH, padW, padH) - stride 1 only (all the path uses; models.lua:646)
local Conv = class('nn.SpatialConvolution', Gemm)
function Conv:__init(nIn, nOut, kW, kH, dW, dH, padW, padH)
   Module.__init(self)
   assert((dW or 1) == 1 and (dH or 1) == 1, 'the G/D definitions only use stride 1')
   self.nInputPlane, self.nOutputPlane, self.kW, self.kH, self.dW, self.dH = nIn, nOut, kW, kH, 1, 1
   self.padW = padW or 0; self.padH = padH or self.padW
   self.weight = Device.new(nOut, nIn, kH, kW); self.bias = Device.new(nOut)
   self.gradWeight = Device.new(nOut, nIn, kH, kW):zero(); self.gradBias = Device.new(nOut):zero()
   self:reset()
end
function Conv:_wdims() return self.nOutputPlane, self.nInputPlane, self.kH, self.kW end
function Conv:_fan_in() return self.kW * self.kH * self.nInputPlane end
function Conv:_geom(x)    -- N, low-res H, W, output H, W
   local N, Cc, H, W = x.shape[1], x.shape[2], x.shape[3], x.shape[4]
   assert(Cc == self.nInputPlane, tostring(self) .. ': got ' .. Cc .. ' input planes')
   local div = (x.ups == 1) and 2 or 1
   return N, H / div, W / div, H + 2 * self.padH - self.kH + 1, W + 2 * self.padW - self.kW + 1
end
function Conv:_can_fold_ups() return self.kH == self.kW and self.kH % 2 == 1 and self.padH == (self.kH - 1) / 2 and self.padW == self.padH end
Conv.winograd, Conv.winograd_min_tiles = true, 2048
function Conv:_use_wino(x)   -- F(2x2,3x3) on the phase convolutions of upsample2 -> conv5x5 (csrc/winograd.hip)
   if not (Conv.winograd and x.ups == 1 and self.kH == 5 and self.kW == 5 and self.padH == 2 and self.padW == 2) then return false end
   local N, Hp, Wp = self:_geom(x)
   if N * Hp * Wp / 4 < Conv.winograd_min_tiles then return false end
   local ok = tonumber(C.cg_conv2d_ups2_wino_supported(N, Hp, Wp, self.nInputPlane, self.nOutputPlane, 5, 2)) == 1
   if ok and not self._wino then self._wino = true; self._ph_version = nil end
   return ok
end
function Conv:updateOutput(input)
   local x = to_device(input):nhwc(true)
   if x.ups == 1 and not self:_can_fold_ups() then x = x:materialise() end
   local N, Hp, Wp, Ho, Wo = self:_geom(x)
   local out = self:_buf('out', { N, self.nOutputPlane, Ho, Wo }, 'nhwc')
   if self:_use_wino(x) then
      self:_ensure_packed_ups()
      local v = self:_buf('wino_v', { tonumber(C.cg_conv2d_ups2_wino_v_floats(N, Hp, Wp, self.nInputPlane)) })
      check(C.cg_conv2d_ups2_wino_forward(S(), x.ptr, self._u_fwd.ptr, self.bias.ptr, out.ptr, v.ptr, N, Hp, Wp, self.nInputPlane, self.nOutputPlane))
   elseif x.ups == 1 then
      self:_ensure_packed_ups()
      conv_forward(x, self._wf_ph.ptr, self.bias.ptr, out, N, Hp, Wp, self.nInputPlane, self.nOutputPlane, self.kH, self.kW, self.padH, self.padW, 1)
   else
      self:_ensure_packed()
      conv_forward(x, self._wf.ptr, self.bias.ptr, out, N, Hp, Wp, self.nInputPlane, self.nOutputPlane, self.kH, self.kW, self.padH, self.padW, 0)
   end
   self._x, self.output = x, out
   return out
end
function Conv:updateGradInput(input, gradOutput)
   local x, dy = self._x, gradOutput:nhwc()
   local N, Hp, Wp = self:_geom(x)
   if x.ups == 1 then
      -- gradient w.r.t. the LOW-RES tensor behind the virtual upsampling (its 2x2 block sum folded in), handed to
      -- nn.SpatialUpSamplingNearest as the dual of its lazy output: logical shape, ups = 1
      local lo = self:_buf('gin_lo', { N, self.nInputPlane, Hp, Wp }, 'nhwc')
      if self:_use_wino(x) then
         local vdy = self:_buf('wino_vdy', { tonumber(C.cg_conv2d_ups2_wino_v_floats(N, Hp, Wp, 4 * self.nOutputPlane)) })
         -- below one workgroup per CU: K slices over blockIdx.z + fixed-order sum (csrc/winograd.hip); 0 floats = the unsplit launch
         local npart = tonumber(C.cg_conv2d_ups2_wino_dgrad_part_floats(N, Hp, Wp, self.nInputPlane, self.nOutputPlane))
         if npart > 0 then
            local part = self:_buf('wino_dpart', { npart })
            check(C.cg_conv2d_ups2_wino_dgrad_split(S(), dy.ptr, self._u_bwd.ptr, lo.ptr, vdy.ptr, part.ptr, N, Hp, Wp, self.nInputPlane, self.nOutputPlane))
         else
            check(C.cg_conv2d_ups2_wino_dgrad(S(), dy.ptr, self._u_bwd.ptr, lo.ptr, vdy.ptr, N, Hp, Wp, self.nInputPlane, self.nOutputPlane))
         end
      else
         local ws, wsb = workspace(C.cg_conv2d_dgrad_ups2_workspace_bytes(N, Hp, Wp, self.nInputPlane, self.nOutputPlane, self.kH, self.padH))
         check(C.cg_conv2d_dgrad_ups2(S(), dy.ptr, self._wb_ph.ptr, lo.ptr, N, Hp, Wp, self.nInputPlane, self.nOutputPlane, self.kH, self.padH, ws, wsb))
      end
      self.gradInput = Device.view_of(lo.store, lo.off, { N, self.nInputPlane, 2 * Hp, 2 * Wp }, 'nhwc', 1)
      return self.gradInput
   end
   local Ho, Wo = dy.shape[3], dy.shape[4]
   local gi = self:_buf('gin', { N, self.nInputPlane, x.shape[3], x.shape[4] }, 'nhwc')
   local wb = self._wb and self._wb.ptr or self.weight.ptr
   conv_forward(dy, wb, nil, gi, N, Ho, Wo, self.nOutputPlane, self.nInputPlane, self.kH, self.kW, self.kH - 1 - self.padH, self.kW - 1 - self.padW, 0)
   self.gradInput = gi
   return gi
end
function Conv:accGradParameters(input, gradOutput, scale)
   local x, dy = self._x, gradOutput:nhwc()
   local N, Hp, Wp = self:_geom(x)
   if x.ups == 1 and self:_use_wino(x) then   -- Winograd-domain weight gradient from the V the forward of this batch left behind
      local v = self:_buf('wino_v', { tonumber(C.cg_conv2d_ups2_wino_v_floats(N, Hp, Wp, self.nInputPlane)) })
      local ws, wsb = workspace(C.cg_conv2d_ups2_wino_wgrad_workspace_bytes(N, Hp, Wp, self.nInputPlane, self.nOutputPlane))
      check(C.cg_conv2d_ups2_wino_wgrad(S(), v.ptr, dy.ptr, self.gradWeight.ptr, self.gradBias.ptr, N, Hp, Wp, self.nInputPlane, self.nOutputPlane, scale or 1, ws, wsb))
      return
   end
   local ws, wsb = workspace(C.cg_conv2d_wgrad_workspace_bytes(N, Hp, Wp, self.nInputPlane, self.nOutputPlane, self.kH, self.kW, self.padH, self.padW, x.ups))
   check(C.cg_conv2d_wgrad(S(), x.ptr, dy.ptr, self.gradWeight.ptr, self.gradBias.ptr, N, Hp, Wp, self.nInputPlane, self.nOutputPlane,
                           self.kH, self.kW, self.padH, self.padW, x.ups, scale or 1, ws, wsb))
end
function Conv:__repr()
   return ('%s(%d -> %d, %dx%d, 1,1, %d,%d)'):format(self.__typename, self.nInputPlane, self.nOutputPlane, self.kW, self.kH, self.padW, self.padH)
end
nn.SpatialConvolution = Conv
cudnn.SpatialConvolution = class('cudnn.SpatialConvolution', Conv)   -- the typename matters to weight-init.lua:54

-- layers/SpatialConvolutionUpsample.lua:1-56: conv to nOut*f^2 planes, then the NCHW buffer [N, nOut*f^2, h, w] is
-- REINTERPRETED (a plain view, not a pixel shuffle) as [N, nOut, h*f, w*f]
local ConvUp = class('nn.SpatialConvolutionUpsample', Conv)
function ConvUp:__init(nIn, nOut, kW, kH, factor)
   assert(kW % 2 == 1, 'kW has to be odd'); assert(kH % 2 == 1, 'kH has to be odd')
   self.factor = factor or 2
   self.nOutputPlaneU = nOut
   Conv.__init(self, nIn, nOut * self.factor * self.factor, kW, kH, 1, 1, (kW - 1) / 2, (kH - 1) / 2)
end
function ConvUp:updateOutput(input)
   local y = Conv.updateOutput(self, input):plain()
   self.h, self.w = y.shape[3], y.shape[4]
   self.output = y:view(y.shape[1], self.nOutputPlaneU, self.h * self.factor, self.w * self.factor)
   return self.output
end
function ConvUp:_view_back(g) g = g:plain(); return g:view(g.shape[1], self.nOutputPlaneU * self.factor * self.factor, self.h, self.w) end
function ConvUp:updateGradInput(input, gradOutput) return Conv.updateGradInput(self, input, self:_view_back(gradOutput)) end
function ConvUp:accGradParameters(input, gradOutput, scale) Conv.accGradParameters(self, input, self:_view_back(gradOutput), scale) end
nn.SpatialConvolutionUpsample = ConvUp
cudnn.SpatialConvolutionUpsample = class('cudnn.SpatialConvolutionUpsample', ConvUp)

-- --------------------------------------------------------------------------------------------------- activations
-- nn.PReLU(nil, nil, true) (models.lua:201): the extra arguments are ignored upstream; one shared slope, init 0.25
local PReLU = class('nn.PReLU')
function PReLU:__init() Module.__init(self); self.weight = Device.new(1):fill(0.25); self.gradWeight = Device.new(1):zero() end
function PReLU:updateOutput(input)
   local x = to_device(input):materialise()
   local out = self:_buf('out', x.shape, x.fmt)
   check(C.cg_prelu_forward(S(), x.ptr, self.weight.ptr, out.ptr, x.n))
   self._x, self.output = x, out
   return out
end
function PReLU:_bwd(gradOutput, galpha, scale)
   local x = self._x
   local dy = (x.fmt == 'nhwc') and gradOutput:nhwc() or gradOutput:plain()
   local gi = self:_buf('gin', x.shape, x.fmt)
   local ws, wsb = nil, 0
   if galpha then ws, wsb = workspace(C.cg_prelu_backward_workspace_bytes(x.n)) end
   check(C.cg_prelu_backward(S(), x.ptr, dy.ptr, self.weight.ptr, gi.ptr, galpha, scale, x.n, ws, wsb))
   self.gradInput = gi
   return gi
end
function PReLU:updateGradInput(input, gradOutput) return self:_bwd(gradOutput, nil, 0) end
function PReLU:accGradParameters(input, gradOutput, scale) self:_bwd(gradOutput, self.gradWeight.ptr, scale or 1) end
function PReLU:backward(input, gradOutput, scale) return self:_bwd(gradOutput, self.gradWeight.ptr, scale or 1) end   -- one pass
nn.PReLU = PReLU

-- nn.LeakyReLU(s) (LeakyReLU.lua:5-31): negative_scale 0.333; x == 0 takes the positive branch
local LeakyReLU = class('nn.LeakyReLU')
function LeakyReLU:__init(s) Module.__init(self); self.negative_scale = s or 0.333 end
function LeakyReLU:updateOutput(input)
   local x = to_device(input):materialise()
   local out = self:_buf('out', x.shape, x.fmt)
   check(C.cg_leakyrelu_forward(S(), x.ptr, out.ptr, self.negative_scale, x.n))
   self._x, self.output = x, out
   return out
end
function LeakyReLU:updateGradInput(input, gradOutput)
   local x = self._x
   local dy = (x.fmt == 'nhwc') and gradOutput:nhwc() or gradOutput:plain()
   local gi = self:_buf('gin', x.shape, x.fmt)
   check(C.cg_leakyrelu_backward(S(), x.ptr, dy.ptr, gi.ptr, self.negative_scale, x.n))
   self.gradInput = gi
   return gi
end
nn.LeakyReLU = LeakyReLU

local Sigmoid = class('nn.Sigmoid')
function Sigmoid:updateOutput(input)
   local x = to_device(input):materialise()
   local out = self:_buf('out', x.shape, x.fmt)
   check(C.cg_sigmoid_forward(S(), x.ptr, out.ptr, x.n))
   self.output = out
   return out
end
function Sigmoid:updateGradInput(input, gradOutput)
   local y = self.output
   local dy = (y.fmt == 'nhwc') and gradOutput:nhwc() or gradOutput:plain()
   local gi = self:_buf('gin', y.shape, y.fmt)
   check(C.cg_sigmoid_backward(S(), y.ptr, dy.ptr, gi.ptr, y.n))
   self.gradInput = gi
   return gi
end
nn.Sigmoid = Sigmoid

-- nn.SpatialBatchNormalization(n) [upstream]: eps 1e-5, momentum 0.1, affine, gamma ~ U(0,1), beta 0.
-- The batch statistics are exchanged as fp64 sums, so a data-parallel host all-reduces them between the two calls
-- (nn.sync_bn = function(sums_ptr, count) ... end; see catgan.comm).
local SBN = class('nn.SpatialBatchNormalization')
function SBN:__init(n, eps, momentum)
   Module.__init(self)
   self.nFeature, self.eps, self.momentum = n, eps or 1e-5, momentum or 0.1
   self.weight = Device.new(n); check(C.cg_rng_uniform(S(), self.weight.ptr, n, 0.0, 1.0, rng.seed, rng.take(n)))
   self.bias = Device.new(n):zero()
   self.gradWeight = Device.new(n):zero(); self.gradBias = Device.new(n):zero()
   self.running_mean = Device.new(n):zero(); self.running_var = Device.new(n):fill(1.0)
   self.save_mean = Device.new(n):zero(); self.save_std = Device.new(n):zero()   -- 1/sqrt(var+eps), as THNN's save_std
   local function dbl(k) local p = ffi.new('void*[1]'); check(C.cg_malloc(p, 8 * k)); return ffi.gc(ffi.cast('double*', p[0]), function(q) C.cg_free(q) end) end
   self._sums, self._bsums, self._bsums_g = dbl(2 * n), dbl(2 * n), dbl(2 * n)
end
function SBN:updateOutput(input)
   local x = to_device(input):nhwc()
   local N, Cc, H, W = x.shape[1], x.shape[2], x.shape[3], x.shape[4]
   local M = N * H * W
   local out = self:_buf('out', x.shape, 'nhwc')
   if not self.train then
      check(C.cg_bn_forward_eval(S(), x.ptr, out.ptr, self.weight.ptr, self.bias.ptr, self.running_mean.ptr, self.running_var.ptr, M, Cc, self.eps))
   else
      check(C.cg_bn_stats(S(), x.ptr, M, Cc, self._sums))
      self._count = M
      if nn.sync_bn then self._count = nn.sync_bn(self._sums, 2 * Cc, M) end   -- all-reduce(sum) of the 2C sums, global count
      check(C.cg_bn_forward(S(), x.ptr, out.ptr, self.weight.ptr, self.bias.ptr, self._sums, self._count, M, Cc, self.eps, self.momentum,
                            self.running_mean.ptr, self.running_var.ptr, self.save_mean.ptr, self.save_std.ptr))
   end
   self._x, self.output = x, out
   return out
end
function SBN:_bwd(gradOutput, acc, scale)
   assert(self.train, 'BN backward in evaluate() mode is not on the path')
   local x, dy = self._x, gradOutput:nhwc()
   local N, Cc, H, W = x.shape[1], x.shape[2], x.shape[3], x.shape[4]
   local M = N * H * W
   check(C.cg_bn_backward_stats(S(), x.ptr, dy.ptr, self.save_mean.ptr, self.save_std.ptr, M, Cc, self._bsums))
   local gs = self._bsums
   if nn.sync_bn then
      check(C.cg_memcpy_d2d(S(), self._bsums_g, self._bsums, 16 * Cc)); nn.sync_bn(self._bsums_g, 2 * Cc, M); gs = self._bsums_g
   end
   local gi = self:_buf('gin', x.shape, 'nhwc')
   check(C.cg_bn_backward(S(), x.ptr, dy.ptr, self.weight.ptr, self.save_mean.ptr, self.save_std.ptr, gs, self._count, self._bsums, M, Cc,
                          gi.ptr, acc and self.gradWeight.ptr or nil, acc and self.gradBias.ptr or nil, scale))
   self.gradInput = gi
   return gi
end
function SBN:updateGradInput(input, gradOutput) return self:_bwd(gradOutput, false, 0) end
function SBN:accGradParameters(input, gradOutput, scale) self:_bwd(gradOutput, true, scale or 1) end
function SBN:backward(input, gradOutput, scale) return self:_bwd(gradOutput, true, scale or 1) end
function SBN:__repr() return ('nn.SpatialBatchNormalization(%d)'):format(self.nFeature) end
nn.SpatialBatchNormalization = SBN

-- ----------------------------------------------------------------------------------------- shape / data movement
-- nn.View(...): the logical NCHW reinterpretation; with NHWC storage this is where the permutation lives
local View = class('nn.View')
function View:__init(...) Module.__init(self); self.sizes = { ... } end
function View:updateOutput(input)
   local x = to_device(input):plain()
   local N = x.shape[1]
   self._in_shape = copy_shape(x.shape)
   if #self.sizes == 3 then
      local Cc, H, W = self.sizes[1], self.sizes[2], self.sizes[3]
      local out = self:_buf('out', { N, Cc, H, W }, 'nhwc')
      check(C.cg_nchw_to_nhwc(S(), x.ptr, out.ptr, N, Cc, H, W))
      self.output = out
   else
      local s = { N }; for i, v in ipairs(self.sizes) do s[i + 1] = v end
      self.output = x:view(s)
   end
   return self.output
end
function View:updateGradInput(input, gradOutput)
   self.gradInput = gradOutput:plain():view(self._in_shape)
   return self.gradInput
end
nn.View = View

-- nn.Copy(intype, outtype, forceCopy, dontCast) (models.lua:643,704; utils/nn_utils.lua:638-643): the host <-> device boundary
local Copy = class('nn.Copy')
function Copy:__init(intype, outtype) Module.__init(self); self.intype, self.outtype = intype or 'torch.FloatTensor', outtype or 'torch.FloatTensor' end
local function to_type(x, typ)
   if typ:find('Cuda') then return to_device(x) end
   if x.__typename == 'torch.CudaTensor' then return x:float() end
   return x
end
function Copy:updateOutput(input) self.output = to_type(input, self.outtype); return self.output end
function Copy:updateGradInput(input, gradOutput) self.gradInput = to_type(gradOutput, self.intype); return self.gradInput end
nn.Copy = Copy

-- nn.Transpose({a,b},...): only the NCHW <-> BHWD pair the spatial transformer uses (models.lua:870,903); with NHWC
-- storage both are relabelings of the same memory
local Transpose = class('nn.Transpose')
function Transpose:__init(...) Module.__init(self); self.permutations = { ... } end
local function transpose_apply(x, perms, reverse)
   local order = { 1, 2, 3, 4 }
   local a, b, step = 1, #perms, 1
   if reverse then a, b, step = #perms, 1, -1 end
   for i = a, b, step do local p = perms[i]; order[p[1]], order[p[2]] = order[p[2]], order[p[1]] end
   local key = table.concat(order, '')
   if key == '1342' then      -- NCHW -> BHWD
      x = to_device(x):nhwc()
      return Device.view_of(x.store, x.off, { x.shape[1], x.shape[3], x.shape[4], x.shape[2] }, 'plain', 0)
   elseif key == '1423' then  -- BHWD -> NCHW
      assert(x.fmt == 'plain')
      return Device.view_of(x.store, x.off, { x.shape[1], x.shape[4], x.shape[2], x.shape[3] }, 'nhwc', 0)
   end
   error('nn.Transpose: permutation ' .. key .. ' is not on the path')
end
function Transpose:updateOutput(input) self.output = transpose_apply(input, self.permutations, false); return self.output end
function Transpose:updateGradInput(input, gradOutput) self.gradInput = transpose_apply(gradOutput, self.permutations, true); return self.gradInput end
nn.Transpose = Transpose

-- nn.SpatialUpSamplingNearest(2): never materialised when a convolution consumes it (ups flag)
local Up = class('nn.SpatialUpSamplingNearest')
function Up:__init(s) Module.__init(self); assert(s == 2, 'the generators only use scale 2'); self.scale_factor = s end
function Up:updateOutput(input)
   local x = to_device(input):nhwc()
   self.output = Device.view_of(x.store, x.off, { x.shape[1], x.shape[2], 2 * x.shape[3], 2 * x.shape[4] }, 'nhwc', 1)
   return self.output
end
function Up:updateGradInput(input, gradOutput)
   if gradOutput.fmt == 'nhwc' and gradOutput.ups == 1 then   -- the consumer convolution already folded the 2x2 block sum
      local g = gradOutput
      self.gradInput = Device.view_of(g.store, g.off, { g.shape[1], g.shape[2], g.shape[3] / 2, g.shape[4] / 2 }, 'nhwc', 0)
      return self.gradInput
   end
   local g = gradOutput:nhwc()
   local N, Cc, H2, W2 = g.shape[1], g.shape[2], g.shape[3], g.shape[4]
   local gi = self:_buf('gin', { N, Cc, H2 / 2, W2 / 2 }, 'nhwc')
   check(C.cg_upsample2x_backward(S(), g.ptr, gi.ptr, N, H2 / 2, W2 / 2, Cc))
   self.gradInput = gi
   return gi
end
nn.SpatialUpSamplingNearest = Up

local function pool_class(name, fwd, bwd)
   local P = class(name)
   function P:__init(kW, kH, dW, dH)
      Module.__init(self)
      dW, dH = dW or kW, dH or kH
      assert(kW == 2 and kH == 2 and dW == 2 and dH == 2, 'the path only pools 2x2 stride 2')
   end
   function P:updateOutput(input)
      local x = to_device(input):nhwc()
      local N, Cc, H, W = x.shape[1], x.shape[2], x.shape[3], x.shape[4]
      local out = self:_buf('out', { N, Cc, H / 2, W / 2 }, 'nhwc')
      check(fwd(S(), x.ptr, out.ptr, N, H, W, Cc))
      self._x, self.output = x, out
      return out
   end
   function P:updateGradInput(input, gradOutput)
      local x, g = self._x, gradOutput:nhwc()
      local gi = self:_buf('gin', x.shape, 'nhwc')
      check(bwd(S(), x, g, gi))
      self.gradInput = gi
      return gi
   end
   return P
end
nn.SpatialAveragePooling = pool_class('nn.SpatialAveragePooling', C.cg_avgpool2_forward,
   function(s, x, g, gi) return C.cg_avgpool2_backward(s, g.ptr, gi.ptr, x.shape[1], x.shape[3], x.shape[4], x.shape[2]) end)
nn.SpatialMaxPooling = pool_class('nn.SpatialMaxPooling', C.cg_maxpool2_forward,
   function(s, x, g, gi) return C.cg_maxpool2_backward(s, x.ptr, g.ptr, gi.ptr, x.shape[1], x.shape[3], x.shape[4], x.shape[2]) end)

-- nn.SpatialDropout(p) [upstream, era]: train y = x * mask[n,c] (no rescale); evaluate y = (1-p) x
local SDrop = class('nn.SpatialDropout')
function SDrop:__init(p) Module.__init(self); self.p = p or 0.5 end
function SDrop:updateOutput(input)
   local x = to_device(input):nhwc()
   local N, Cc, H, W = x.shape[1], x.shape[2], x.shape[3], x.shape[4]
   local out = self:_buf('out', x.shape, 'nhwc')
   if self.train then
      self.noise = self:_buf('noise', { N, Cc })
      check(C.cg_rng_bernoulli(S(), self.noise.ptr, N * Cc, 1.0 - self.p, 1.0, rng.seed, rng.take(N * Cc)))
      check(C.cg_mask_mul(S(), x.ptr, self.noise.ptr, out.ptr, N, H * W, Cc, 1))
   else
      out:copy(x):mul(1.0 - self.p)
   end
   self.output = out
   return out
end
function SDrop:updateGradInput(input, gradOutput)
   local g = gradOutput:nhwc()
   local N, Cc, H, W = g.shape[1], g.shape[2], g.shape[3], g.shape[4]
   local gi = self:_buf('gin', g.shape, 'nhwc')
   if self.train then check(C.cg_mask_mul(S(), g.ptr, self.noise.ptr, gi.ptr, N, H * W, Cc, 1))
   else gi:copy(g):mul(1.0 - self.p) end
   self.gradInput = gi
   return gi
end
nn.SpatialDropout = SDrop

-- nn.Dropout(p) v2 [upstream]: train y = x * mask / (1-p); evaluate identity
local Drop = class('nn.Dropout')
function Drop:__init(p) Module.__init(self); self.p = p or 0.5 end
function Drop:updateOutput(input)
   local x = to_device(input):materialise()
   if not self.train then self.output = x; return x end
   self.noise = self:_buf('noise', x.shape, x.fmt)
   check(C.cg_rng_bernoulli(S(), self.noise.ptr, x.n, 1.0 - self.p, 1.0 / (1.0 - self.p), rng.seed, rng.take(x.n)))
   local out = self:_buf('out', x.shape, x.fmt)
   check(C.cg_mask_mul(S(), x.ptr, self.noise.ptr, out.ptr, 1, x.n, 1, 0))
   self.output = out
   return out
end
function Drop:updateGradInput(input, gradOutput)
   if not self.train then self.gradInput = gradOutput; return gradOutput end
   local g = gradOutput
   local gi = self:_buf('gin', g.shape, g.fmt)
   check(C.cg_mask_mul(S(), g.ptr, self.noise.ptr, gi.ptr, 1, g.n, 1, 0))
   self.gradInput = gi
   return gi
end
nn.Dropout = Drop

-- ------------------------------------------------------------------------------------- spatial transformer (stn)
local ATMG = class('nn.AffineTransformMatrixGenerator')
function ATMG:__init(rot, scale, trans) Module.__init(self); self.useRotation, self.useScale, self.useTranslation = rot and 1 or 0, scale and 1 or 0, trans and 1 or 0 end
function ATMG:updateOutput(input)
   local p = to_device(input):plain()
   local out = self:_buf('out', { p.shape[1], 2, 3 })
   check(C.cg_affine_matrix_forward(S(), p.ptr, out.ptr, p.shape[1], self.useRotation, self.useScale, self.useTranslation))
   self._p, self.output = p, out
   return out
end
function ATMG:updateGradInput(input, gradOutput)
   local p = self._p
   local gi = self:_buf('gin', p.shape)
   check(C.cg_affine_matrix_backward(S(), p.ptr, gradOutput.ptr, gi.ptr, p.shape[1], self.useRotation, self.useScale, self.useTranslation))
   self.gradInput = gi
   return gi
end
nn.AffineTransformMatrixGenerator = ATMG

local AGG = class('nn.AffineGridGeneratorBHWD')
function AGG:__init(h, w) Module.__init(self); self.height, self.width = h, w end
function AGG:updateOutput(input)
   local Tm = to_device(input)
   local out = self:_buf('out', { Tm.shape[1], self.height, self.width, 2 })
   check(C.cg_affine_grid_forward(S(), Tm.ptr, out.ptr, Tm.shape[1], self.height, self.width))
   self.output = out
   return out
end
function AGG:updateGradInput(input, gradOutput)
   local N = gradOutput.shape[1]
   local gi = self:_buf('gin', { N, 2, 3 })
   check(C.cg_affine_grid_backward(S(), gradOutput.ptr, gi.ptr, N, self.height, self.width))
   self.gradInput = gi
   return gi
end
nn.AffineGridGeneratorBHWD = AGG

-- input = { images [N,H,W,C], grids [N,h,w,2] }.  The reference pins this module to the CPU even in GPU mode
-- (models.lua:889-899) because stn's scatter was non-reproducible; here it is a deterministic device kernel, so
-- `sampler:type(...)` stays the no-op the reference patches in.
local Sampler = class('nn.BilinearSamplerBHWD')
function Sampler:updateOutput(input)
   local img, grid = input[1], input[2]
   local N, Hi, Wi, Cc = img.shape[1], img.shape[2], img.shape[3], img.shape[4]
   local Ho, Wo = grid.shape[2], grid.shape[3]
   local out = self:_buf('out', { N, Ho, Wo, Cc })
   check(C.cg_bilinear_sampler_forward(S(), img.ptr, grid.ptr, out.ptr, N, Hi, Wi, Cc, Ho, Wo))
   self.output = out
   return out
end
function Sampler:updateGradInput(input, gradOutput)
   local img, grid = input[1], input[2]
   local N, Hi, Wi, Cc = img.shape[1], img.shape[2], img.shape[3], img.shape[4]
   local Ho, Wo = grid.shape[2], grid.shape[3]
   local gimg, ggrid = self:_buf('gimg', img.shape), self:_buf('ggrid', grid.shape)
   check(C.cg_bilinear_sampler_backward(S(), img.ptr, grid.ptr, gradOutput.ptr, gimg.ptr, ggrid.ptr, N, Hi, Wi, Cc, Ho, Wo))
   self.gradInput = { gimg, ggrid }
   return self.gradInput
end
nn.BilinearSamplerBHWD = Sampler

-- ------------------------------------------------------------------------------------------------------ criterion
-- nn.BCECriterion() (train.lua:181): sizeAverage, eps 1e-12 [upstream].  adversarial.lua hands it host tensors (the
-- discriminator ends in an nn.Copy back to the host, models.lua:704) and reads the loss as a number.
local BCE = {}
BCE.__index = BCE
function nn.BCECriterion() return setmetatable({ output = 0, gradInput = nil, __typename = 'nn.BCECriterion' }, BCE) end
function BCE:forward(input, target)
   local x, t = to_device(input):plain(), to_device(target)
   assert(x.n == t.n)
   self._loss = self._loss or Device.new(1)
   check(C.cg_bce_forward(S(), x.ptr, t.ptr, self._loss.ptr, x.n))
   local h = ffi.new('float[1]')
   check(C.cg_memcpy_d2h(S(), h, self._loss.ptr, 4)); check(C.cg_stream_sync(S()))
   self.output = h[0]
   return self.output
end
function BCE:backward(input, target)
   local host = input.__typename == 'torch.FloatTensor'
   local x, t = to_device(input):plain(), to_device(target)
   self._g = self._g or Device.new(x.shape)
   check(C.cg_bce_backward(S(), x.ptr, t.ptr, self._g.ptr, x.n))
   self.gradInput = host and self._g:float() or self._g
   return self.gradInput
end

-- ------------------------------------------------------------------------------------ the validator's two extra classes
-- create_V32 / create_V16 (models.lua:763-799) end in Linear -> nn.BatchNormalization -> ... -> nn.SoftMax.  V is outside the
-- hot-path scope (it is trained by train_v.lua), but the UNCHANGED train.lua loads one (train.lua:119-123), puts it in evaluate()
-- mode and utils/nn_utils.lua:686-711 (rateWithV) runs it forward once per epoch.  So: inference only, on the host, on [N, n] tensors.
local BN1 = class('nn.BatchNormalization')
function BN1:__init(n, eps, momentum, affine)
   Module.__init(self)
   self.eps, self.momentum = eps or 1e-5, momentum or 0.1
   self.running_mean, self.running_var = Host.new(n):zero(), Host.new(n):fill(1)
   if affine ~= false then self.weight, self.bias = Host.new(n):uniform(0, 1), Host.new(n):zero(); self.gradWeight, self.gradBias = Host.new(n):zero(), Host.new(n):zero() end
end
function BN1:updateOutput(input)
   assert(not self.train, 'nn.BatchNormalization: training mode is outside the hot-path scope (V is trained by train_v.lua); call :evaluate()')
   local dev = input.__typename == 'torch.CudaTensor'
   local x = dev and input:float() or input
   local N, n = x.shape[1], x.shape[2]
   local out = Host.new(N, n)
   local xd, od = x:data(), out:data()
   local mean, var = self.running_mean:float():data(), self.running_var:float():data()
   local w, b = self.weight and self.weight:float():data(), self.bias and self.bias:float():data()
   for i = 0, N - 1 do
      for j = 0, n - 1 do
         local v = (xd[i * n + j] - mean[j]) / math.sqrt(var[j] + self.eps)
         od[i * n + j] = w and v * w[j] + b[j] or v
      end
   end
   self.output = dev and out:cuda() or out
   return self.output
end
function BN1:updateGradInput() error('nn.BatchNormalization: backward is outside the hot-path scope') end
nn.BatchNormalization = BN1

local SoftMax = class('nn.SoftMax')
function SoftMax:updateOutput(input)
   local dev = input.__typename == 'torch.CudaTensor'
   local x = dev and input:float() or input
   local n = x.shape[#x.shape]
   local rows = x.n / n
   local out = Host.new(x.shape)
   local xd, od = x:data(), out:data()
   for i = 0, rows - 1 do
      local mx = -math.huge
      for j = 0, n - 1 do mx = math.max(mx, xd[i * n + j]) end
      local sum = 0
      for j = 0, n - 1 do local e = math.exp(xd[i * n + j] - mx); od[i * n + j] = e; sum = sum + e end
      for j = 0, n - 1 do od[i * n + j] = od[i * n + j] / sum end
   end
   self.output = dev and out:cuda() or out
   return self.output
end
function SoftMax:updateGradInput() error('nn.SoftMax: backward is outside the hot-path scope') end
nn.SoftMax = SoftMax

return { nn = nn, cudnn = cudnn }
