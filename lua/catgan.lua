--[[ catgan.lua — LuaJIT FFI binding of include/catgan.h and the pattern for re-creating Torch7's nn classes on it.

NOT EXECUTED in the build container (no Lua of any kind there; SURVEY.md Appendix C).  It is the declarative twin of
cat-generator_amd/{tensor,nn,optim}.py: same C ABI, same call sequence per module method.  The Python layer is the
one the tests drive; this file is what a maintainer drops next to models.lua so that

    nn = require 'catgan'.nn ; cudnn = require 'catgan'.cudnn ; optim = require 'catgan'.optim

makes models.lua:196-228 / :640-711 and adversarial.lua build and step on the engine unchanged.
]]
local ffi = require 'ffi'

-- 1. cdef straight from the header: strip comments / preprocessor lines / the extern "C" guard
local function load_header(path)
   local src = assert(io.open(path)):read('*a')
   src = src:gsub('/%*.-%*/', ''):gsub('//[^\n]*', ''):gsub('\n%s*#[^\n]*', '\n')
   src = src:gsub('extern%s+"C"%s*{', ''):gsub('\n}%s*\n', '\n')
   ffi.cdef(src)
end
load_header(os.getenv('CATGAN_HEADER') or 'include/catgan.h')
local C = ffi.load(os.getenv('CATGAN_LIB') or 'cat-generator_amd/lib/libcatgan_hip.so')

local function check(rc) if rc ~= 0 then error(ffi.string(C.cg_last_error()), 2) end end

local M = { nn = {}, cudnn = {}, optim = {}, C = C }
local stream = nil  -- default HIP stream; cg_stream_create() for a private one

-- 2. device tensor: opaque pointer + logical (Torch7) shape + physical format ('plain' | 'nhwc') + ups flag
local Tensor = {}; Tensor.__index = Tensor
function Tensor.new(shape, fmt)
   local n = 1; for _, s in ipairs(shape) do n = n * s end
   local p = ffi.new('void*[1]'); check(C.cg_malloc(p, n * 4))
   local t = setmetatable({ ptr = ffi.gc(ffi.cast('float*', p[0]), C.cg_free), shape = shape, fmt = fmt or 'plain',
                            ups = 0, n = n }, Tensor)
   return t
end
function Tensor:zero() check(C.cg_memset_zero(stream, self.ptr, self.n * 4)); return self end
function Tensor:fill(v) check(C.cg_fill(stream, self.ptr, v, self.n)); return self end
function Tensor:copy(src)  -- src: catgan tensor (same layout) or a host float* / torch.FloatTensor:data()
   if getmetatable(src) == Tensor then check(C.cg_memcpy_d2d(stream, self.ptr, src.ptr, self.n * 4))
   else check(C.cg_memcpy_h2d(stream, self.ptr, src, self.n * 4)) end
   return self
end
function Tensor:clamp(lo, hi) check(C.cg_clamp(stream, self.ptr, lo, hi, self.n)); return self end
function Tensor:add(alpha, other) if not other then alpha, other = 1, alpha end
   check(C.cg_axpy(stream, alpha, other.ptr, self.ptr, self.n)); return self end
function Tensor:mul(a) check(C.cg_scale(stream, self.ptr, a, self.n)); return self end
function Tensor:nElement() return self.n end
function Tensor:size(i) return i and self.shape[i] or self.shape end
M.Tensor = Tensor

local scratch = { ptr = nil, bytes = 0 }
local function workspace(bytes)
   bytes = math.max(tonumber(bytes), 4096)
   if scratch.bytes < bytes then
      local p = ffi.new('void*[1]'); check(C.cg_malloc(p, bytes))
      scratch.ptr, scratch.bytes = ffi.gc(p[0], C.cg_free), bytes
   end
   return scratch.ptr, scratch.bytes
end

-- 3. nn.Module protocol (LeakyReLU.lua:5-31 and layers/SpatialConvolutionUpsample.lua show the upstream shape of it)
local Module = {}; Module.__index = Module
function Module:forward(input) return self:updateOutput(input) end
function Module:backward(input, gradOutput, scale)
   self:updateGradInput(input, gradOutput); self:accGradParameters(input, gradOutput, scale or 1)
   return self.gradInput
end
function Module:accGradParameters() end
local function class(name, parent)
   local c = setmetatable({ __typename = name }, { __index = parent or Module,
      __call = function(cls, ...) local o = setmetatable({}, cls); o:__init(...); return o end })
   c.__index = c
   return c
end

-- nn.SpatialConvolution(nIn, nOut, kW, kH, dW, dH, padW, padH)  (models.lua:646; cudnn.* at :206)
local Conv = class('nn.SpatialConvolution')
function Conv:__init(nIn, nOut, kW, kH, dW, dH, padW, padH)
   assert((dW or 1) == 1 and (dH or 1) == 1, 'stride 1 only')
   self.nInputPlane, self.nOutputPlane, self.kW, self.kH = nIn, nOut, kW, kH
   self.padW = padW or 0; self.padH = padH or self.padW
   self.weight = Tensor.new({ nOut, nIn, kH, kW }); self.bias = Tensor.new({ nOut })
   self.gradWeight = Tensor.new({ nOut, nIn, kH, kW }):zero(); self.gradBias = Tensor.new({ nOut }):zero()
   self.wf = Tensor.new({ kH * kW * nIn, nOut }); self.wb = Tensor.new({ kH * kW * nOut, nIn })
   self:reset()
end
function Conv:pack()  -- after every parameter update (the Python layer tracks a mutation counter instead)
   check(C.cg_pack_conv_weight(stream, self.weight.ptr, self.wf.ptr, self.wb.ptr, self.nOutputPlane, self.nInputPlane, self.kH, self.kW))
end
function Conv:updateOutput(x)  -- x: NHWC, possibly with ups = 1 (virtual nearest upsampling)
   local N, H, W = x.shape[1], x.shape[3], x.shape[4]
   local Hp, Wp = bit.rshift(H, x.ups), bit.rshift(W, x.ups)
   local Ho, Wo = H + 2 * self.padH - self.kH + 1, W + 2 * self.padW - self.kW + 1
   self.output = self.output or Tensor.new({ N, self.nOutputPlane, Ho, Wo }, 'nhwc')
   local ws, wsb = workspace(C.cg_conv2d_workspace_bytes(N, Hp, Wp, self.nInputPlane, self.nOutputPlane, self.kH, self.kW, self.padH, self.padW, x.ups))
   self:pack()
   check(C.cg_conv2d_forward(stream, x.ptr, self.wf.ptr, self.bias.ptr, self.output.ptr, N, Hp, Wp,
                             self.nInputPlane, self.nOutputPlane, self.kH, self.kW, self.padH, self.padW, x.ups, ws, wsb))
   self._x = x
   return self.output
end
function Conv:updateGradInput(x, dy)
   local N, Ho, Wo = dy.shape[1], dy.shape[3], dy.shape[4]
   self.gradInput = self.gradInput or Tensor.new({ N, self.nInputPlane, x.shape[3], x.shape[4] }, 'nhwc')
   local pH, pW = self.kH - 1 - self.padH, self.kW - 1 - self.padW
   local ws, wsb = workspace(C.cg_conv2d_workspace_bytes(N, Ho, Wo, self.nOutputPlane, self.nInputPlane, self.kH, self.kW, pH, pW, 0))
   check(C.cg_conv2d_forward(stream, dy.ptr, self.wb.ptr, nil, self.gradInput.ptr, N, Ho, Wo,
                             self.nOutputPlane, self.nInputPlane, self.kH, self.kW, pH, pW, 0, ws, wsb))
   return self.gradInput
end
function Conv:accGradParameters(x, dy, scale)
   local N, H, W = x.shape[1], x.shape[3], x.shape[4]
   local Hp, Wp = bit.rshift(H, x.ups), bit.rshift(W, x.ups)
   local ws, wsb = workspace(C.cg_conv2d_wgrad_workspace_bytes(N, Hp, Wp, self.nInputPlane, self.nOutputPlane, self.kH, self.kW, self.padH, self.padW, x.ups))
   check(C.cg_conv2d_wgrad(stream, x.ptr, dy.ptr, self.gradWeight.ptr, self.gradBias.ptr, N, Hp, Wp, self.nInputPlane,
                           self.nOutputPlane, self.kH, self.kW, self.padH, self.padW, x.ups, scale or 1, ws, wsb))
end
function Conv:reset(stdv)  -- host-side init exactly as nn.SpatialConvolution:reset [upstream]
   stdv = stdv and stdv * math.sqrt(3) or 1 / math.sqrt(self.kW * self.kH * self.nInputPlane)
   local n = self.weight.n; local h = ffi.new('float[?]', n)
   for i = 0, n - 1 do h[i] = (math.random() * 2 - 1) * stdv end
   self.weight:copy(h)
   local b = ffi.new('float[?]', self.nOutputPlane)
   for i = 0, self.nOutputPlane - 1 do b[i] = (math.random() * 2 - 1) * stdv end
   self.bias:copy(b)
end
M.nn.SpatialConvolution = Conv
M.cudnn.SpatialConvolution = class('cudnn.SpatialConvolution', Conv)  -- typename matters to weight-init.lua:54

-- nn.SpatialUpSamplingNearest(2): never materialised — a relabeled handle with ups = 1
local Up = class('nn.SpatialUpSamplingNearest')
function Up:__init(s) assert(s == 2) end
function Up:updateOutput(x)
   self.output = setmetatable({ ptr = x.ptr, shape = { x.shape[1], x.shape[2], 2 * x.shape[3], 2 * x.shape[4] },
                                fmt = 'nhwc', ups = 1, n = x.n * 4 }, Tensor)
   return self.output
end
function Up:updateGradInput(x, dy)
   self.gradInput = self.gradInput or Tensor.new(x.shape, 'nhwc')
   check(C.cg_upsample2x_backward(stream, dy.ptr, self.gradInput.ptr, x.shape[1], x.shape[3], x.shape[4], x.shape[2]))
   return self.gradInput
end
M.nn.SpatialUpSamplingNearest = Up

-- nn.PReLU(nil, nil, true)
local PReLU = class('nn.PReLU')
function PReLU:__init() self.weight = Tensor.new({ 1 }):fill(0.25); self.gradWeight = Tensor.new({ 1 }):zero() end
function PReLU:updateOutput(x)
   self.output = self.output or Tensor.new(x.shape, x.fmt)
   check(C.cg_prelu_forward(stream, x.ptr, self.weight.ptr, self.output.ptr, x.n)); return self.output
end
function PReLU:backward(x, dy, scale)
   self.gradInput = self.gradInput or Tensor.new(x.shape, x.fmt)
   local ws, wsb = workspace(C.cg_prelu_backward_workspace_bytes(x.n))
   check(C.cg_prelu_backward(stream, x.ptr, dy.ptr, self.weight.ptr, self.gradInput.ptr, self.gradWeight.ptr, scale or 1, x.n, ws, wsb))
   return self.gradInput
end
M.nn.PReLU = PReLU

-- The remaining classes follow the same three-line pattern, one C call per method; the Python twin lists them:
--   nn.Linear, nn.View, nn.SpatialBatchNormalization (cg_bn_stats -> [all-reduce] -> cg_bn_forward; backward likewise),
--   nn.Sigmoid, nn.LeakyReLU, nn.SpatialAveragePooling, nn.SpatialMaxPooling, nn.SpatialDropout, nn.Dropout,
--   nn.Concat, nn.ConcatTable, nn.Sequential, nn.Copy, nn.Transpose, nn.AffineTransformMatrixGenerator,
--   nn.AffineGridGeneratorBHWD, nn.BilinearSamplerBHWD, nn.SpatialConvolutionUpsample, nn.BCECriterion.

-- 4. optim.adam(opfunc, x, state): one fused kernel on the flat vectors (adversarial.lua:245,262)
function M.optim.adam(opfunc, x, config, state)
   config = config or {}; state = state or config
   local fx, dfdx = opfunc(x)
   state.t = (state.t or 0) + 1
   state.m = state.m or Tensor.new(x.shape):zero(); state.v = state.v or Tensor.new(x.shape):zero()
   check(C.cg_adam_step(stream, x.ptr, dfdx.ptr, state.m.ptr, state.v.ptr, x.n, config.learningRate or 1e-3,
                        config.beta1 or 0.9, config.beta2 or 0.999, config.epsilon or 1e-8, state.t,
                        config.l1 or 0, config.l2 or 0, config.clamp or 0, 1))
   return x, { fx }
end

return M
