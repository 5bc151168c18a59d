--[[ catgan.torchx — the rest of the `torch` table and of the tensor methods that the reference's host files call OUTSIDE the training
step (train.lua:54-57,119-142,260; utils/nn_utils.lua:9,164,205,262-274,392-421; adversarial.lua:319-338; dataset.lua:161): plain
host-side Lua over torch.FloatTensor.  Loaded by lua/torch.lua after catgan.tensor. ]]
local ffi = require 'ffi'
local T = require 'catgan.tensor'
local Host, Device, torch = T.Host, T.Device, T.torch
local t7 = require 'catgan.t7'

torch.save, torch.load = t7.save, t7.load
local nthreads = 1
function torch.setnumthreads(n) nthreads = n end        -- the engine's host side is one thread per GPU (SURVEY.md 8b)
function torch.getnumthreads() return nthreads end
function torch.floor(x) return math.floor(x) end        -- adversarial.lua:330 applies them to numbers
function torch.sqrt(x) if type(x) == 'number' then return math.sqrt(x) end; return x:clone():sqrt() end
function torch.clamp(t, lo, hi) return t:clone():clamp(lo, hi) end
function torch.repeatTensor(t, ...) return t:repeatTensor(...) end
function torch.randperm(n)                               -- dataset.lua:161: Fisher-Yates on math.random (seeded by train.lua:63)
   local p = Host.new(n)
   local d = p:data()
   for i = 0, n - 1 do d[i] = i + 1 end
   for i = n - 1, 1, -1 do local j = math.random(0, i); d[i], d[j] = d[j], d[i] end
   return p
end
function torch.cat(a, b, dim)
   assert(dim == nil or dim == 1, 'torch.cat: first dimension only')
   local s = T.copy_shape(a.shape); s[1] = a.shape[1] + b.shape[1]
   local o = Host.new(s)
   ffi.copy(o:data(), a:data(), a.n * 4); ffi.copy(o:data() + a.n, b:data(), b.n * 4)
   return o
end

-- host tensor methods
function Host:clamp(lo, hi) local d = self:data(); for i = 0, self.n - 1 do d[i] = math.max(lo, math.min(hi, d[i])) end; return self end
function Host:sqrt() local d = self:data(); for i = 0, self.n - 1 do d[i] = math.sqrt(d[i]) end; return self end
function Host:div(a) return self:mul(1 / a) end
function Host:ne(o)          -- x:ne(x):sum() > 0 is the reference's NaN test (utils/nn_utils.lua:164)
   local r = Host.new(self.shape)
   local a, b, d = self:data(), type(o) == 'number' and nil or o:data(), r:data()
   for i = 0, self.n - 1 do d[i] = (a[i] ~= (b and b[i] or o)) and 1 or 0 end
   return r
end
function Host:randn(...)     -- weights:randn(weights:size()) (utils/nn_utils.lua:9): Box-Muller on math.random
   local d = self:data()
   for i = 0, self.n - 1, 2 do
      local u1, u2 = 1 - math.random(), math.random()
      local m = math.sqrt(-2 * math.log(u1))
      d[i] = m * math.cos(2 * math.pi * u2)
      if i + 1 < self.n then d[i + 1] = m * math.sin(2 * math.pi * u2) end
   end
   return self
end
function Host:select(dim, idx)   -- first dimension only: a view (utils/nn_utils.lua:262-264 takes the colour planes)
   assert(dim == 1, 'select: first dimension only')
   return self[idx]
end
function Host:sub(a, b) return self[{ { a, b } }] end       -- rows a..b of the first dimension
function Host:typeAs(o) if o.__typename == 'torch.CudaTensor' then return self:cuda() end; return self end
function Host:repeatTensor(...)
   local reps = { ... }
   local nd = #reps
   local shape = {}
   for i = 1, nd - #self.shape do shape[i] = 1 end
   for i = 1, #self.shape do shape[#shape + 1] = self.shape[i] end
   local out_shape = {}
   for i = 1, nd do out_shape[i] = shape[i] * reps[i] end
   local out = Host.new(out_shape)
   local s, d = self:data(), out:data()
   local idx = {}
   for i = 1, nd do idx[i] = 0 end
   for o = 0, out.n - 1 do
      local src, mul = 0, 1
      for i = nd, 1, -1 do src = src + (idx[i] % shape[i]) * mul; mul = mul * shape[i] end
      d[o] = s[src]
      for i = nd, 1, -1 do idx[i] = idx[i] + 1; if idx[i] < out_shape[i] then break end; idx[i] = 0 end
   end
   return out
end
function Device:typeAs(o) if o.__typename == 'torch.FloatTensor' then return self:float() end; return self end
function Device:sum() return self:float():sum() end
function Device:ne(o) return self:float():ne(type(o) == 'number' and o or o:float()) end
function Device:select(dim, idx) return self:float():select(dim, idx) end

return torch
