--[[ catgan.optim — torch/optim's functions on flat device vectors (adversarial.lua:240-248, 257-265) and
optim.ConfusionMatrix for the binary case of adversarial.lua:74,101-106,285-289.

optim.adam follows torch/optim's form [upstream]: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
x -= lr sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)   (eps added to sqrt(v) without bias-correcting v).
One kernel per update (cg_adam_step / cg_sgd_step / cg_adagrad_step) instead of ~8 elementwise passes.  The penalty and
the clamp of adversarial.lua:92-98,110-112 stay where the reference puts them (inside feval, as tensor methods); a host
that wants them fused passes config.fused = { l1 =, l2 =, clamp = } as the Python host does. ]]
local abi = require 'catgan.ffi'
local T = require 'catgan.tensor'
local C, check = abi.C, abi.check
local Device = T.Device

local optim = {}

local function fused(config)
   local f = config.fused
   if f then return f.l1 or 0, f.l2 or 0, f.clamp or 0, 1 end
   return 0, 0, 0, 0
end

function optim.adam(opfunc, x, config, state)
   config = config or {}; state = state or config
   local lr, b1, b2, eps = config.learningRate or 0.001, config.beta1 or 0.9, config.beta2 or 0.999, config.epsilon or 1e-8
   local fx, dfdx = opfunc(x)
   if fx == false then return x, { fx } end   -- the accuracy gate's `return false,false` (adversarial.lua:165): skip the update
   if not state.t then
      state.t = 0
      state.m = Device.new(x.shape):zero(); state.v = Device.new(x.shape):zero()
   end
   state.t = state.t + 1
   local l1, l2, clamp, wb = fused(config)
   check(C.cg_adam_step(T.stream, x.ptr, dfdx.ptr, state.m.ptr, state.v.ptr, x.n, lr, b1, b2, eps, state.t, l1, l2, clamp, wb))
   x:touch()
   return x, { fx }
end

-- optim.sgd as train.lua:201-204 configures it (learningRate, momentum; Torch7 defaults: dampening = momentum, no
-- Nesterov, no weight decay).  First step with momentum: v = g [upstream: dfdx:clone()].
function optim.sgd(opfunc, x, config, state)
   config = config or {}; state = state or config
   local lr, mom = config.learningRate or 1e-3, config.momentum or 0
   local damp = config.dampening or mom
   local fx, dfdx = opfunc(x)
   if fx == false then return x, { fx } end
   local first = state.dfdx == nil
   if mom ~= 0 and first then state.dfdx = Device.new(x.shape):zero() end
   state.evalCounter = (state.evalCounter or 0) + 1
   local l1, l2, clamp, wb = fused(config)
   check(C.cg_sgd_step(T.stream, x.ptr, dfdx.ptr, mom ~= 0 and state.dfdx.ptr or nil, x.n, lr, mom,
                       (mom ~= 0 and first) and 0 or damp, l1, l2, clamp, wb))
   x:touch()
   return x, { fx }
end

-- optim.adagrad (train.lua:193-196): paramVariance += g^2; x -= lr * g / (sqrt(paramVariance) + 1e-10)
function optim.adagrad(opfunc, x, config, state)
   config = config or {}; state = state or config
   local lr = config.learningRate or 1e-3
   local fx, dfdx = opfunc(x)
   if fx == false then return x, { fx } end
   state.paramVariance = state.paramVariance or Device.new(x.shape):zero()
   state.evalCounter = (state.evalCounter or 0) + 1
   local l1, l2, clamp, wb = fused(config)
   check(C.cg_adagrad_step(T.stream, x.ptr, dfdx.ptr, state.paramVariance.ptr, x.n, lr, l1, l2, clamp, wb))
   x:touch()
   return x, { fx }
end

-- optim.ConfusionMatrix(classes): :zero() :add(prediction, target) (1-based class indices, adversarial.lua:104)
-- :updateValids() .totalValid and a printable form (adversarial.lua:285-289)
local CM = {}
CM.__index = CM
function optim.ConfusionMatrix(classes)
   local n = type(classes) == 'table' and #classes or classes
   local o = setmetatable({ nclasses = n, classes = classes, mat = {}, totalValid = 0, averageValid = 0 }, CM)
   return o:zero()
end
function CM:zero()
   for i = 1, self.nclasses do self.mat[i] = {}; for j = 1, self.nclasses do self.mat[i][j] = 0 end end
   self.totalValid, self.averageValid = 0, 0
   return self
end
function CM:add(prediction, target) self.mat[target][prediction] = self.mat[target][prediction] + 1 end
function CM:updateValids()
   local diag, total = 0, 0
   for i = 1, self.nclasses do for j = 1, self.nclasses do total = total + self.mat[i][j]; if i == j then diag = diag + self.mat[i][j] end end end
   self.totalValid = total > 0 and diag / total or 0
end
CM.__tostring = function(self)
   self:updateValids()
   local s = { 'ConfusionMatrix:' }
   for i = 1, self.nclasses do s[#s + 1] = '[' .. table.concat(self.mat[i], ' ') .. ']' end
   s[#s + 1] = (' + global correct: %.4f%%'):format(100 * self.totalValid)
   return table.concat(s, '\n')
end

return optim
