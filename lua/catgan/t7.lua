--[[ catgan.t7 — torch.save / torch.load in Torch7's binary serialisation (train.lua:119-142 reads {V = ...}, {D, G, optstate, epoch, ...};
train.lua:260 writes {D, G, opt, plot_data, epoch, normalize_mean, normalize_std}).  The Lua twin of cat-generator_amd/t7.py -
same byte layout, little endian, 8-byte longs:
    object    := int32 type, payload
    type 0 nil | 1 number: float64 | 2 string: int32 length, bytes | 5 boolean: int32
    type 3 table : int32 index, [first time:] int32 count, count x (key object, value object)
    type 4 torch : int32 index, [first time:] string "V 1", string class name, class payload
    tensor  payload : int32 nDim, nDim x int64 size, nDim x int64 stride, int64 storageOffset (1-based), storage object
    storage payload : int64 count, count x element
    other classes   : one table object holding the instance's fields (File.lua's default)
Tensors that are views of one storage (Module:getParameters' flat vector) share ONE storage object in the file, as in Torch7.
Device tensors are written as torch.CudaTensor / torch.CudaStorage (cutorch's names) in logical (plain) order and come back on
the device.  PARITY UNPINNED, like t7.py: no Torch7-written file exists in the build image (tests/test_t7.py pins the Python
twin to hand-assembled bytes; scripts/check_lua_binding.py holds this file's tags and class names to t7.py's). ]]
local ffi = require 'ffi'
local T = require 'catgan.tensor'
local Host, Device = T.Host, T.Device

local M = {}
local TYPE_NIL, TYPE_NUMBER, TYPE_STRING, TYPE_TABLE, TYPE_TORCH, TYPE_BOOLEAN = 0, 1, 2, 3, 4, 5

-- class registry: typename -> class table (torch.factory's role); filled from nn / cudnn on first use
local function class_of(name)
   local cg = require 'catgan'
   local ns, cls = name:match('^(%w+)%.(.+)$')
   local tab = ({ nn = cg.nn, cudnn = cg.cudnn, optim = cg.optim })[ns]
   return tab and tab[cls] or nil
end

-- ------------------------------------------------------------------------------------------------ writer
local Writer = {}
Writer.__index = Writer
local i32, i64, f64 = ffi.new('int32_t[1]'), ffi.new('int64_t[1]'), ffi.new('double[1]')
function Writer:int(v) i32[0] = v; self.f:write(ffi.string(i32, 4)) end
function Writer:long(v) i64[0] = v; self.f:write(ffi.string(i64, 8)) end
function Writer:double(v) f64[0] = v; self.f:write(ffi.string(f64, 8)) end
function Writer:string(s) self:int(#s); self.f:write(s) end
function Writer:ref(obj)       -- writes the index; true if the object was written before
   local idx = self.index[obj]
   if idx then self:int(idx); return true end
   self.next = self.next + 1
   self.index[obj] = self.next
   self:int(self.next)
   return false
end
function Writer:header(name) self:string('V 1'); self:string(name) end

function Writer:storage(store, cuda, host_copy)
   self:int(TYPE_TORCH)
   if self:ref(store) then return end
   self:header(cuda and 'torch.CudaStorage' or 'torch.FloatStorage')
   self:long(store.n)
   self.f:write(ffi.string(host_copy or store.data, store.n * 4))
end

function Writer:tensor(t)
   self:int(TYPE_TORCH)
   if self:ref(t) then return end
   local cuda = t.__typename == 'torch.CudaTensor'
   if cuda and (t.fmt ~= 'plain' or t.ups ~= 0) then t = t:plain() end   -- the logical Torch7 layout
   self:header(cuda and 'torch.CudaTensor' or 'torch.FloatTensor')
   local shape = t.shape
   if t.n == 0 or #shape == 0 then self:int(0); self:long(1); self:int(TYPE_NIL); return end
   self:int(#shape)
   for i = 1, #shape do self:long(shape[i]) end
   local strides, s = {}, 1
   for i = #shape, 1, -1 do strides[i] = s; s = s * shape[i] end
   for i = 1, #shape do self:long(strides[i]) end
   self:long(t.off + 1)
   if cuda then
      local copy = self.host_copies[t.store]
      if not copy then                                   -- one device -> host copy per storage
         copy = ffi.new('float[?]', math.max(t.store.n, 1))
         local abi = require 'catgan.ffi'
         abi.check(abi.C.cg_memcpy_d2h(T.stream, copy, t.store.ptr, t.store.n * 4)); abi.check(abi.C.cg_stream_sync(T.stream))
         self.host_copies[t.store] = copy
      end
      self:storage(t.store, true, copy)
   else
      self:storage(t.store, false, nil)
   end
end

function Writer:object(o)
   local ty = type(o)
   if o == nil then self:int(TYPE_NIL)
   elseif ty == 'boolean' then self:int(TYPE_BOOLEAN); self:int(o and 1 or 0)
   elseif ty == 'number' then self:int(TYPE_NUMBER); self:double(o)
   elseif ty == 'string' then self:int(TYPE_STRING); self:string(o)
   elseif ty == 'table' then
      if o.__tensor then return self:tensor(o) end
      local mt = getmetatable(o)
      local name = mt and mt.__typename
      if name then                                       -- an instance of a torch class: its fields as one table
         self:int(TYPE_TORCH)
         if self:ref(o) then return end
         self:header(name)
         local fields = {}
         for k, v in pairs(o) do
            if type(v) ~= 'function' and k ~= '_bufs' and k ~= '_pnet' and k ~= '_pnet_n' then fields[k] = v end   -- engine-side caches stay out
         end
         return self:object(fields)
      end
      self:int(TYPE_TABLE)
      if self:ref(o) then return end
      local n = 0
      for _, v in pairs(o) do if type(v) ~= 'function' then n = n + 1 end end
      self:int(n)
      for k, v in pairs(o) do
         if type(v) ~= 'function' then self:object(k); self:object(v) end
      end
   else
      error('torch.save: cannot serialise a ' .. ty .. ' (functions are not part of the checkpoint format here)')
   end
end

function M.save(path, obj)
   local f = assert(io.open(path, 'wb'), 'torch.save: cannot write ' .. tostring(path))
   local w = setmetatable({ f = f, index = {}, next = 0, host_copies = {} }, Writer)
   w:object(obj)
   f:close()
   return path
end

-- ------------------------------------------------------------------------------------------------ reader
local Reader = {}
Reader.__index = Reader
function Reader:bytes(n)
   local s = self.f:read(n)
   if n > 0 and (not s or #s ~= n) then error('torch.load: truncated file') end
   return s or ''
end
function Reader:int() return ffi.cast('const int32_t*', self:bytes(4))[0] end
function Reader:long() return tonumber(ffi.cast('const int64_t*', self:bytes(8))[0]) end
function Reader:double() return ffi.cast('const double*', self:bytes(8))[0] end
function Reader:string() return self:bytes(self:int()) end

local STORAGES = { ['torch.FloatStorage'] = 'host', ['torch.CudaStorage'] = 'device' }
local TENSORS = { ['torch.FloatTensor'] = 'host', ['torch.CudaTensor'] = 'device' }

function Reader:object()
   local ty = self:int()
   if ty == TYPE_NIL then return nil
   elseif ty == TYPE_NUMBER then return self:double()
   elseif ty == TYPE_STRING then return self:string()
   elseif ty == TYPE_BOOLEAN then return self:int() ~= 0
   elseif ty == TYPE_TABLE then
      local idx = self:int()
      if self.objects[idx] ~= nil then return self.objects[idx] end
      local out = {}
      self.objects[idx] = out
      for _ = 1, self:int() do
         local k = self:object()
         out[k] = self:object()
      end
      return out
   elseif ty == TYPE_TORCH then
      local idx = self:int()
      if self.objects[idx] ~= nil then return self.objects[idx] end
      local version = self:string()
      local name = version:sub(1, 2) == 'V ' and self:string() or version      -- pre-versioning files: the name comes first
      if STORAGES[name] then
         local n = self:long()
         local raw = self:bytes(n * 4)
         local st
         if STORAGES[name] == 'host' then
            st = Host.new(math.max(n, 1)).store; st.n = n
            ffi.copy(st.data, raw, n * 4)
         else
            st = Device.new(math.max(n, 1)).store; st.n = n
            local abi = require 'catgan.ffi'
            abi.check(abi.C.cg_memcpy_h2d(T.stream, st.ptr, raw, n * 4)); abi.check(abi.C.cg_stream_sync(T.stream))
         end
         self.objects[idx] = st
         return st
      elseif TENSORS[name] then
         local nd = self:int()
         local size, stride = {}, {}
         for i = 1, nd do size[i] = self:long() end
         for i = 1, nd do stride[i] = self:long() end
         local off = self:long() - 1
         local store = self:object()
         local t
         if store == nil or nd == 0 then
            t = TENSORS[name] == 'host' and Host.new(0) or Device.new(0)
         else
            local s = 1                                   -- the engine's tensors are contiguous views
            for i = nd, 1, -1 do assert(stride[i] == s or size[i] == 1, 'torch.load: non-contiguous tensor'); s = s * size[i] end
            if TENSORS[name] == 'host' then t = Host.new(0); t.store, t.off, t.shape, t.n = store, off, size, T.prod(size)
            else t = Device.view_of(store, off, size, 'plain', 0) end
         end
         self.objects[idx] = t
         return t
      elseif name:match('^torch%.%a+Storage$') or name:match('^torch%.%a+Tensor$') then
         error('torch.load: ' .. name .. ' is not a type the engine holds (FloatTensor / CudaTensor only)')
      end
      local cls = class_of(name)
      local obj = cls and setmetatable({}, cls) or { __typename = name }
      self.objects[idx] = obj
      local fields = self:object()
      if type(fields) == 'table' then for k, v in pairs(fields) do rawset(obj, k, v) end end
      if cls and obj._bufs == nil then obj._bufs = {} end
      return obj
   end
   error('torch.load: unknown object type ' .. tostring(ty) .. ' (functions and other types are not supported)')
end

function M.load(path)
   local f = assert(io.open(path, 'rb'), 'torch.load: cannot open ' .. tostring(path))
   local r = setmetatable({ f = f, objects = {} }, Reader)
   local obj = r:object()
   f:close()
   return obj
end

return M
