--[[ catgan.net — host side of the planned executor (cg_net_*, include/catgan.h; csrc/net.hip), the LuaJIT twin of
cat-generator_amd/planned.py.

adversarial.lua calls MODEL_D:forward / :backward and MODEL_G:forward / :backward on whole networks (:84-89, :182-197).  Those
calls do not walk the module tree in Lua: the tree is DESCRIBED to the library once - one cg_net_add per module, with the
arguments models.lua passed to the constructor - and every pass is ONE cg_net_forward / cg_net_backward call.  Segment
fusion, the lockstep execution of D32_st3's three identical branches, the side stream, deferred weight-gradient reductions,
weight re-packing and the data-parallel exchanges are planned below the C ABI, so this file only marshals: module tree ->
builder calls, parameter tensors -> cg_net_bind, tensors in / out.  tools/abi_replay replays exactly this call sequence
(recorded from the Python twin at the benchmarked batch) without any interpreter: tests/test_abi_step.py.

The per-module classes of catgan.nn stay what a module used on its own gets (the nn.Module protocol). ]]
local ffi = require 'ffi'
local abi = require 'catgan.ffi'
local T = require 'catgan.tensor'
local C, check = abi.C, abi.check
local Device = T.Device

local KIND = {
   ['nn.Sequential'] = 0, ['nn.Concat'] = 1, ['nn.ConcatTable'] = 2, ['nn.Linear'] = 3, ['nn.SpatialConvolution'] = 4,
   ['cudnn.SpatialConvolution'] = 4, ['nn.PReLU'] = 5, ['nn.LeakyReLU'] = 6, ['nn.Sigmoid'] = 7, ['nn.SpatialBatchNormalization'] = 8,
   ['nn.View'] = 9, ['nn.Copy'] = 10, ['nn.Transpose'] = 11, ['nn.SpatialUpSamplingNearest'] = 12, ['nn.SpatialAveragePooling'] = 13,
   ['nn.SpatialMaxPooling'] = 14, ['nn.SpatialDropout'] = 15, ['nn.Dropout'] = 16, ['nn.AffineTransformMatrixGenerator'] = 17,
   ['nn.AffineGridGeneratorBHWD'] = 18, ['nn.BilinearSamplerBHWD'] = 19,
}

-- (kind, iargs, fargs) of one module: the arguments of its reference constructor; nil = no entry in the planned executor
local function describe(m)
   local tn = m.__typename
   local k = KIND[tn]
   if not k then return nil end
   if k == 1 then return k, { m.dimension }, {} end
   if k == 3 then return k, { m.weight.shape[2], m.weight.shape[1] }, {} end
   if k == 4 then return k, { m.nInputPlane, m.nOutputPlane, m.kW, m.kH, m.padW, m.padH, m.dW, m.dH }, {} end
   if k == 6 then return k, {}, { m.negative_scale } end
   if k == 8 then return k, { m.nFeature }, { m.eps, m.momentum } end
   if k == 9 then
      if #m.sizes ~= 1 and #m.sizes ~= 3 then return nil end
      return k, m.sizes, {}
   end
   if k == 11 then
      local order = { 1, 2, 3, 4 }
      for _, p in ipairs(m.permutations) do order[p[1]], order[p[2]] = order[p[2]], order[p[1]] end
      local key = table.concat(order, '')
      if key == '1342' then return k, { 0 }, {} end    -- NCHW -> BHWD (models.lua:870)
      if key == '1423' then return k, { 1 }, {} end    -- BHWD -> NCHW (models.lua:903)
      return nil
   end
   if k == 12 then return k, { m.scale_factor }, {} end
   if k == 15 or k == 16 then return k, {}, { m.p } end
   if k == 17 then return k, { m.useRotation, m.useScale, m.useTranslation }, {} end
   if k == 18 then return k, { m.height, m.width }, {} end
   return k, {}, {}
end

local Net = {}
Net.__index = Net

local function add(self, m, parent)
   local kind, ia, fa = describe(m)
   if kind == nil then return false end
   local ia_c = ffi.new('long[8]'); for i, v in ipairs(ia) do ia_c[i - 1] = v end
   local fa_c = ffi.new('float[4]'); for i, v in ipairs(fa) do fa_c[i - 1] = v end
   local id = ffi.new('int[1]')
   check(C.cg_net_add(self.h, parent, kind, ia_c, #ia, fa_c, #fa, id))
   self.ids[m] = id[0]
   self.mods[#self.mods + 1] = m
   if m.modules then
      for _, c in ipairs(m.modules) do if not add(self, c, id[0]) then return false end end
   end
   return true
end

-- Net.new(root): nil when the tree holds a module the planned executor has no entry for (the container then walks its modules)
function Net.new(root)
   local hp = ffi.new('void*[1]')
   check(C.cg_net_create(hp))
   local self = setmetatable({ h = ffi.gc(hp[0], function(h) C.cg_net_destroy(h) end), ids = {}, mods = {}, root = root }, Net)
   if not add(self, root, -1) then return nil end
   return self
end

-- what the host may have changed since the last pass: parameter tensors re-pointed by getParameters() (train.lua:184-185) or
-- replaced (models.lua:860), parameters moved by optim.adam (a storage's version counter), training() / evaluate()
function Net:sync()
   local sig, version = {}, 0
   local seen = {}
   for _, m in ipairs(self.mods) do
      for slot, names in ipairs({ { 'weight', 'gradWeight' }, { 'bias', 'gradBias' } }) do
         local w, g = m[names[1]], m[names[2]]
         if w ~= nil and g ~= nil and w.ptr ~= nil then
            sig[#sig + 1] = tostring(w.ptr) .. tostring(g.ptr)
            if not seen[w.store] then seen[w.store] = true; version = version + w.store.version end
         end
      end
      if m.running_mean then sig[#sig + 1] = tostring(m.running_mean.ptr) end
      sig[#sig + 1] = m.train and 't' or 'e'
   end
   sig = table.concat(sig, ',')
   if sig ~= self._sig then
      for _, m in ipairs(self.mods) do
         local id = self.ids[m]
         if m.weight and m.gradWeight then check(C.cg_net_bind(self.h, id, 0, m.weight.ptr, m.gradWeight.ptr)) end
         if m.bias and m.gradBias then check(C.cg_net_bind(self.h, id, 1, m.bias.ptr, m.gradBias.ptr)) end
         if m.running_mean then check(C.cg_net_bind(self.h, id, 2, m.running_mean.ptr, m.running_var.ptr)) end
         local tn = m.__typename
         if tn == 'nn.SpatialBatchNormalization' or tn == 'nn.SpatialDropout' or tn == 'nn.Dropout' then
            check(C.cg_net_set_training(self.h, id, m.train and 1 or 0))
         end
      end
      self._sig, self._version = sig, nil
   end
   if version ~= self._version then check(C.cg_net_params_changed(self.h)); self._version = version end
   local comm = package.loaded['catgan.comm']
   if comm and comm.nranks and comm.nranks > 1 and not self._dp then   -- data parallelism: the plan carries the exchanges (SURVEY.md 8e)
      check(C.cg_net_set_dp(self.h, comm.nranks, 1, comm.bn, comm.grad, self.root._bucket_overlap and 1 or 0))
      self._dp = true
   end
end

local function wrap(ptr, nd, dims, fmt)   -- a tensor the plan owns, as a CudaTensor view (not freed by Lua)
   local shape, n = {}, 1
   for i = 1, nd do shape[i] = tonumber(dims[i - 1]); n = n * shape[i] end
   local store = { ptr = ffi.cast('float*', ptr), n = n, version = 0 }
   local ups = (fmt >= 2) and 1 or 0
   return Device.view_of(store, 0, shape, (fmt % 2 == 1) and 'nhwc' or 'plain', ups)
end

-- MODEL:forward(input).  rng: the host's counter-stream position { seed, offset }; the pass's dropout masks are drawn at
-- offset, offset + 1, ... in module order and the offset is advanced by what the pass consumed.
function Net:forward(x, rng, rng_base)
   self:sync()
   local dims = ffi.new('long[4]'); for i, v in ipairs(x.shape) do dims[i - 1] = v end
   local y, ynd, ydims, yfmt, draws = ffi.new('float*[1]'), ffi.new('int[1]'), ffi.new('long[4]'), ffi.new('int[1]'), ffi.new('uint64_t[1]')
   check(C.cg_net_forward(self.h, T.stream, x.ptr, #x.shape, dims, (x.fmt == 'nhwc') and 1 or 0, rng.seed, rng.offset, rng_base, draws,
                          y, ynd, ydims, yfmt))
   rng.offset = rng.offset + tonumber(draws[0])
   self._x = x
   return wrap(y[0], ynd[0], ydims, yfmt[0])
end

-- MODEL:forwardPair(input, input2) / MODEL:pairJoin() (cg_net_forward_pair / cg_net_pair_join, round 6): forward(input) on the step's stream
-- and forward(input2) beside it on a library stream of another hardware queue - adversarial.lua:232-233 (fake images, N/2 rows) and :185
-- (the G-step's pass, N rows) read the same parameters.  Returns the first output; pair_join returns the second, which backward continues.
function Net:forward_pair(x, x2, rng, rng_base)
   self:sync()
   local d1 = ffi.new('long[4]'); for i, v in ipairs(x.shape) do d1[i - 1] = v end
   local d2 = ffi.new('long[4]'); for i, v in ipairs(x2.shape) do d2[i - 1] = v end
   local y, ynd, ydims, yfmt, draws = ffi.new('float*[1]'), ffi.new('int[1]'), ffi.new('long[4]'), ffi.new('int[1]'), ffi.new('uint64_t[1]')
   check(C.cg_net_forward_pair(self.h, T.stream, x.ptr, #x.shape, d1, (x.fmt == 'nhwc') and 1 or 0, x2.ptr, #x2.shape, d2,
                               (x2.fmt == 'nhwc') and 1 or 0, rng.seed, rng.offset, rng_base, draws, y, ynd, ydims, yfmt))
   rng.offset = rng.offset + tonumber(draws[0])
   self._x = x2
   return wrap(y[0], ynd[0], ydims, yfmt[0])
end
function Net:pair_join()
   local y, ynd, ydims, yfmt = ffi.new('float*[1]'), ffi.new('int[1]'), ffi.new('long[4]'), ffi.new('int[1]')
   check(C.cg_net_pair_join(self.h, T.stream, y, ynd, ydims, yfmt))
   return wrap(y[0], ynd[0], ydims, yfmt[0])
end

-- MODEL:backward(input, gradOutput, scale) (acc = true) / MODEL:updateGradInput(input, gradOutput) (acc = false: what
-- fevalG_on_D needs of D, adversarial.lua:192-193 - the reference accumulates D's weight gradients there and never reads them)
function Net:backward(gy, acc, scale)
   local gx, gnd, gdims, gfmt = ffi.new('float*[1]'), ffi.new('int[1]'), ffi.new('long[4]'), ffi.new('int[1]')
   check(C.cg_net_backward(self.h, T.stream, self._x.ptr, gy.ptr, (gy.fmt == 'nhwc') and 1 or 0, acc and 1 or 0, scale or 1, gx, gnd,
                           gdims, gfmt))
   return wrap(gx[0], gnd[0], gdims, gfmt[0])
end

-- join the gradient-bucket all-reduces cg_net_backward started (device-side wait); returns how many buckets travelled
function Net:finish_buckets(comm)
   local n = ffi.new('int[1]')
   check(C.cg_net_buckets(self.h, n))
   if n[0] > 0 and comm and comm.grad then check(C.cg_comm_wait(comm.grad, T.stream)) end
   return n[0]
end

function Net:set_option(name, value) check(C.cg_net_set_option(self.h, name, value)) end

-- Whole-iteration replay: graph = net.capture(function() ... one adversarial.train iteration ... end) records every launch the
-- closure makes on the current stream (cg_graph_begin / _end); graph:launch() replays it.
local Graph = {}
Graph.__index = Graph
function Net.capture(body)
   check(C.cg_graph_begin(T.stream))
   local ok, err = pcall(body)
   local ex = ffi.new('void*[1]')
   check(C.cg_graph_end(T.stream, ex))
   if not ok then error(err, 2) end
   return setmetatable({ h = ffi.gc(ex[0], function(h) C.cg_graph_destroy(h) end) }, Graph)
end
function Graph:launch() check(C.cg_graph_launch(self.h, T.stream)) end

return { Net = Net, describe = describe, KIND = KIND }
