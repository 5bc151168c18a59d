--[[ catgan.tensor — the slice of Torch7's tensor API the hot path touches (SURVEY.md 8b "Tensor API used on the
path"), over LuaJIT FFI buffers.

Two classes, named as the reference names them:
  torch.FloatTensor : host memory (malloc'd fp32, contiguous, row-major).  adversarial.lua:56-62,225-238 builds its batches
                      element by element on these (t[i] = u, t[i][j], t[{{a,b}}]), utils/nn_utils.lua:35-69 fills noise.
  torch.CudaTensor  : device memory behind include/catgan.h.  Logical (Torch7) shape + physical format: 'plain' (row
                      major) or 'nhwc' (feature maps, engine-native) + `ups` (virtual 2x nearest upsampling, consumed by the
                      next convolution).  Every arithmetic method is one C-ABI call on the current stream.
Parameters are views into one flat storage per net (Module:getParameters, train.lua:184-185); a storage carries a version
counter that the mutating methods bump, which is what tells a convolution to re-pack its weights.
]]
local ffi = require 'ffi'
local abi = require 'catgan.ffi'
local C, check = abi.C, abi.check

ffi.cdef [[ void* malloc(size_t); void free(void*); void* memcpy(void*, const void*, size_t); void* memset(void*, int, size_t); ]]

local T = {}
T.stream = nil   -- hipStream_t (NULL = default stream); catgan.set_stream() replaces it

local function prod(shape, from)
   local n = 1
   for i = from or 1, #shape do n = n * shape[i] end
   return n
end
local function copy_shape(s) local r = {}; for i = 1, #s do r[i] = s[i] end; return r end
local function shape_of(...)
   local a = { ... }
   if #a == 1 and type(a[1]) == 'table' and not a[1].__tensor then return copy_shape(a[1]) end
   return a
end

-- ------------------------------------------------------------------------------------------------ host tensor
local Host = { __tensor = true, __typename = 'torch.FloatTensor' }

local function host_view(store, offset, shape)
   return setmetatable({ store = store, off = offset, shape = shape, n = prod(shape) }, Host)
end
function Host.new(...)
   local shape = shape_of(...)
   local n = math.max(prod(shape), 1)
   local p = ffi.cast('float*', ffi.C.malloc(n * 4))
   assert(p ~= nil, 'out of host memory')
   local store = { data = ffi.gc(p, ffi.C.free), n = n }
   return host_view(store, 0, shape)
end
function Host.from_table(tbl)   -- torch.Tensor({0, 1, 0, 0}) (models.lua:860)
   local t = Host.new(#tbl)
   for i = 1, #tbl do t.store.data[i - 1] = tbl[i] end
   return t
end
function Host:data() return self.store.data + self.off end
function Host:type() return 'torch.FloatTensor' end
function Host:float() return self end
function Host:nElement() return self.n end
function Host:dim() return #self.shape end
function Host:size(i) if i then return self.shape[i] end; return self.shape end
function Host:zero() ffi.C.memset(self:data(), 0, self.n * 4); return self end
function Host:fill(v) local d = self:data(); for i = 0, self.n - 1 do d[i] = v end; return self end
function Host:clone() local t = Host.new(self.shape); ffi.C.memcpy(t:data(), self:data(), self.n * 4); return t end
function Host:copy(src)
   assert(src.n == self.n, 'copy: size mismatch')
   if src.__typename == 'torch.CudaTensor' then
      local s = src:plain()   -- logical (NCHW) order
      check(C.cg_memcpy_d2h(T.stream, self:data(), s.ptr, self.n * 4)); check(C.cg_stream_sync(T.stream))
   else
      ffi.C.memcpy(self:data(), src:data(), self.n * 4)
   end
   return self
end
function Host:uniform(a, b) local d = self:data(); for i = 0, self.n - 1 do d[i] = a + (b - a) * math.random() end; return self end
function Host:view(...) local s = shape_of(...); assert(prod(s) == self.n); return host_view(self.store, self.off, s) end
function Host:resizeAs(o) if self.n ~= o.n then local t = Host.new(o.shape); self.store, self.off = t.store, 0 end
   self.shape, self.n = copy_shape(o.shape), o.n; return self end
function Host:mul(a) local d = self:data(); for i = 0, self.n - 1 do d[i] = d[i] * a end; return self end
function Host:add(a, o) if not o then a, o = 1, a end
   local d = self:data()
   if type(o) == 'number' then for i = 0, self.n - 1 do d[i] = d[i] + o end
   else local e = o:data(); for i = 0, self.n - 1 do d[i] = d[i] + a * e[i] end end
   return self end
function Host:sum() local d, s = self:data(), 0; for i = 0, self.n - 1 do s = s + d[i] end; return s end
function Host:cuda() return T.Device.new(self.shape):copy(self) end
-- t[i] -> number (1-D) or sub-tensor view; t[{ {a,b}, {}, ... }] -> narrowed view along the first dimension
function Host.__index(self, k)
   if type(k) == 'number' then
      if #self.shape == 1 then return self.store.data[self.off + k - 1] end
      local sub = {}; for i = 2, #self.shape do sub[i - 1] = self.shape[i] end
      return host_view(self.store, self.off + (k - 1) * prod(self.shape, 2), sub)
   elseif type(k) == 'table' then
      local r = k[1]
      local a, b = 1, self.shape[1]
      if type(r) == 'table' and #r == 2 then a, b = r[1], r[2] elseif type(r) == 'number' then a, b = r, r end
      local s = copy_shape(self.shape); s[1] = b - a + 1
      return host_view(self.store, self.off + (a - 1) * prod(self.shape, 2), s)
   end
   return Host[k]
end
function Host.__newindex(self, k, v)
   if type(k) == 'number' or type(k) == 'table' then
      if type(k) == 'number' and #self.shape == 1 then self.store.data[self.off + k - 1] = v; return end
      local dst = Host.__index(self, k)
      if type(v) == 'number' then dst:fill(v) else dst:copy(v) end
   else
      rawset(self, k, v)
   end
end
Host.__tostring = function(self) return 'torch.FloatTensor of size ' .. table.concat(self.shape, 'x') end
T.Host = Host

-- ---------------------------------------------------------------------------------------------- device tensor
local Device = { __tensor = true, __typename = 'torch.CudaTensor' }
Device.__index = Device

local function dev_storage(nfloats)
   local p = ffi.new('void*[1]')
   check(C.cg_malloc(p, math.max(nfloats, 1) * 4))
   return { ptr = ffi.gc(ffi.cast('float*', p[0]), function(q) C.cg_free(q) end), n = nfloats, version = 0 }
end
local function dev_view(store, off, shape, fmt, ups)
   return setmetatable({ store = store, off = off, ptr = store.ptr + off, shape = shape, n = prod(shape), fmt = fmt or 'plain',
                         ups = ups or 0 }, Device)
end
function Device.new(...)
   local shape = shape_of(...)
   return dev_view(dev_storage(prod(shape)), 0, shape, 'plain', 0)
end
function Device.new_nhwc(shape) local t = Device.new(shape); t.fmt = 'nhwc'; return t end
function Device.view_of(store, off, shape, fmt, ups) return dev_view(store, off, copy_shape(shape), fmt, ups) end
function Device:type() return 'torch.CudaTensor' end
function Device:cuda() return self end
function Device:nElement() return self.n end
function Device:dim() return #self.shape end
function Device:size(i) if i then return self.shape[i] end; return self.shape end
function Device:phys_n() return self.ups == 1 and self.n / 4 or self.n end   -- floats actually stored
function Device:touch() self.store.version = self.store.version + 1; return self end
function Device:zero() check(C.cg_memset_zero(T.stream, self.ptr, self:phys_n() * 4)); return self:touch() end
function Device:fill(v) check(C.cg_fill(T.stream, self.ptr, v, self:phys_n())); return self:touch() end
function Device:clamp(lo, hi) check(C.cg_clamp(T.stream, self.ptr, lo, hi, self.n)); return self:touch() end
function Device:mul(a) check(C.cg_scale(T.stream, self.ptr, a, self.n)); return self:touch() end
function Device:add(a, o)   -- x:add(y) / x:add(alpha, y)  (adversarial.lua:97)
   if not o then a, o = 1, a end
   check(C.cg_axpy(T.stream, a, o.ptr, self.ptr, self.n)); return self:touch()
end
function Device:sign()      -- torch.sign(p): fresh tensor (adversarial.lua:97)
   local r = Device.new(self.shape):zero()
   check(C.cg_axpy_sign(T.stream, 1.0, self.ptr, r.ptr, self.n)); return r
end
function Device:norm(p)     -- torch.norm(p, 1 | 2) (adversarial.lua:94-95)
   local acc = ffi.new('void*[1]'); check(C.cg_malloc(acc, 8))
   local d = ffi.cast('double*', acc[0])
   if p == 1 then check(C.cg_sumabs(T.stream, self.ptr, self.n, d)) else check(C.cg_sumsq(T.stream, self.ptr, self.n, d)) end
   local h = ffi.new('double[1]')
   check(C.cg_memcpy_d2h(T.stream, h, d, 8)); check(C.cg_stream_sync(T.stream)); check(C.cg_free(d))
   return p == 1 and h[0] or math.sqrt(h[0])
end
function Device:clone()
   local t = dev_view(dev_storage(self:phys_n()), 0, copy_shape(self.shape), self.fmt, self.ups)
   check(C.cg_memcpy_d2d(T.stream, t.ptr, self.ptr, self:phys_n() * 4)); return t
end
function Device:copy(src)
   if src.__typename == 'torch.FloatTensor' then   -- host (logical order) -> device in this tensor's physical format
      assert(src.n == self.n, 'copy: size mismatch')
      if self.fmt == 'nhwc' and #self.shape == 4 then
         local tmp = Device.new(self.shape); check(C.cg_memcpy_h2d(T.stream, tmp.ptr, src:data(), self.n * 4))
         check(C.cg_nchw_to_nhwc(T.stream, tmp.ptr, self.ptr, self.shape[1], self.shape[2], self.shape[3], self.shape[4]))
         check(C.cg_stream_sync(T.stream))   -- tmp (and the borrowed host buffer) may go away
      else
         check(C.cg_memcpy_h2d(T.stream, self.ptr, src:data(), self.n * 4)); check(C.cg_stream_sync(T.stream))
      end
   else
      assert(src:phys_n() == self:phys_n() and src.fmt == self.fmt, 'copy: layout mismatch')
      check(C.cg_memcpy_d2d(T.stream, self.ptr, src.ptr, self:phys_n() * 4))
   end
   return self:touch()
end
function Device:float() return Host.new(self.shape):copy(self) end   -- device -> host, logical order
function Device:view(...)      -- plain tensors only (nn.View handles the NHWC <-> NCHW reinterpretation)
   local s = shape_of(...); assert(self.fmt == 'plain' and prod(s) == self.n)
   return dev_view(self.store, self.off, s, 'plain', 0)
end
-- the layouts a module may ask for ------------------------------------------------------------------------------
function Device:materialise()   -- resolve a virtual 2x upsampling
   if self.ups == 0 then return self end
   local N, Cc, H, W = self.shape[1], self.shape[2], self.shape[3], self.shape[4]
   local out = Device.new_nhwc({ N, Cc, H, W })
   check(C.cg_upsample2x_forward(T.stream, self.ptr, out.ptr, N, H / 2, W / 2, Cc)); return out
end
function Device:nhwc(keep_ups)
   if self.fmt == 'nhwc' then if keep_ups or self.ups == 0 then return self end; return self:materialise() end
   assert(#self.shape == 4, 'expected a 4-D feature map')
   local out = Device.new_nhwc(self.shape)
   check(C.cg_nchw_to_nhwc(T.stream, self.ptr, out.ptr, self.shape[1], self.shape[2], self.shape[3], self.shape[4])); return out
end
function Device:plain()
   if self.fmt == 'plain' then return self end
   local x = self:materialise()
   local out = Device.new(x.shape)
   check(C.cg_nhwc_to_nchw(T.stream, x.ptr, out.ptr, x.shape[1], x.shape[2], x.shape[3], x.shape[4])); return out
end
-- PARAMETERS:clone():mul(l2) + torch.sign(PARAMETERS):mul(l1)  (adversarial.lua:97): tensor + tensor
Device.__add = function(a, b) return a:clone():add(b) end
Device.__tostring = function(self) return 'torch.CudaTensor of size ' .. table.concat(self.shape, 'x') .. ' (' .. self.fmt .. ')' end
T.Device = Device

-- ------------------------------------------------------------------------------------------------- torch table
local torch = { FloatTensor = Host.new, CudaTensor = Device.new }
function torch.Tensor(...)
   local a = { ... }
   if #a == 1 and type(a[1]) == 'table' and not a[1].__tensor then
      return Host.from_table(a[1])    -- torch.Tensor({0, 1, 0, 0}): a table argument holds VALUES (models.lua:860)
   end
   return Host.new(...)
end
function torch.zeros(...) return Host.new(...):zero() end
function torch.type(o) return type(o) == 'table' and (o.__typename or (getmetatable(o) or {}).__typename) or type(o) end
function torch.norm(t, p) return t:norm(p or 2) end
function torch.sign(t) return t:sign() end
function torch.manualSeed(s) math.randomseed(s) end
function torch.setdefaulttensortype() end
function torch.setnumthreads() end
function torch.isTensor(o) return type(o) == 'table' and o.__tensor == true end
T.torch = torch
T.prod, T.copy_shape = prod, copy_shape
return T
