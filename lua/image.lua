--[[ shim: `require 'image'` (adversarial.lua:4, train.lua:2, dataset.lua:2) - the slice of torch/image the reference calls:
   image.load(path, 3, 'float')          dataset.lua:98,166   -> FloatTensor [3,H,W] in [0,1]
   image.scale(img, w, h)                dataset.lua:99,127   the SAME rule as the loader / its device kernel / the oracle
                                                              (cat-generator_amd/dataset.py:_scale_axis): rows then columns,
                                                              linear + corner-aligned up, box average with weighted ends down
   image.save(path, tensor)              utils/nn_utils.lua:582   PNG (stored deflate blocks) or PPM by extension
   image.rgb2yuv / yuv2rgb / rgb2hsl / hsl2rgb   utils/nn_utils.lua:209-242
   image.toDisplayTensor{input=, padding=, nrow=}, image.display{...}    adversarial.lua:346 (a file under OPT.save, no window)
Host-side, off the training step.  Decoding: binary PPM / PGM natively; JPEG through libturbojpeg, bound with the FFI when it
is there (the reference's dataset is *.jpg, train.lua:87).  Everything works on torch.FloatTensor (catgan.tensor Host). ]]
local ffi = require 'ffi'
local bit = require 'bit'
local T = require 'catgan.tensor'
local Host = T.Host

local image = {}

local function dims(t)   -- [C,H,W] or [H,W]
   local s = t:size()
   if #s == 2 then return 1, s[1], s[2] end
   assert(#s == 3, 'image: expected a [C,H,W] or [H,W] tensor')
   return s[1], s[2], s[3]
end

-- ------------------------------------------------------------------------------------------------ load
local tj = nil
local function turbojpeg()
   if tj ~= nil then return tj or nil end
   local ok, lib = pcall(ffi.load, 'turbojpeg')
   if not ok then tj = false; return nil end
   ffi.cdef [[
      typedef void* tjhandle;
      tjhandle tjInitDecompress(void);
      int tjDecompressHeader3(tjhandle h, const unsigned char* buf, unsigned long size, int* w, int* ht, int* subsamp, int* cs);
      int tjDecompress2(tjhandle h, const unsigned char* buf, unsigned long size, unsigned char* dst, int w, int pitch, int ht, int pixfmt, int flags);
      int tjDestroy(tjhandle h);
   ]]
   tj = lib
   return lib
end

local function from_bytes(px, C, H, W)   -- interleaved 8-bit [H][W][C] -> FloatTensor [C,H,W] / 255
   local t = Host.new(C, H, W)
   local d = t:data()
   for y = 0, H - 1 do
      for x = 0, W - 1 do
         for c = 0, C - 1 do d[(c * H + y) * W + x] = px[(y * W + x) * C + c] / 255 end
      end
   end
   return t
end

local function load_pnm(bytes)
   local magic, w, h, maxv, pos = bytes:match('^(P[56])%s+(%d+)%s+(%d+)%s+(%d+)%s()')
   assert(magic and tonumber(maxv) == 255, 'image.load: only binary 8-bit PPM / PGM')
   local C = magic == 'P6' and 3 or 1
   local px = ffi.cast('const unsigned char*', bytes) + (pos - 1)
   return from_bytes(px, C, tonumber(h), tonumber(w))
end

local function load_jpeg(bytes)
   local lib = turbojpeg()
   assert(lib, 'image.load: JPEG needs libturbojpeg (ffi.load failed); convert the dataset to PPM or install it')
   local h = lib.tjInitDecompress()
   local w, ht, ss, cs = ffi.new('int[1]'), ffi.new('int[1]'), ffi.new('int[1]'), ffi.new('int[1]')
   assert(lib.tjDecompressHeader3(h, bytes, #bytes, w, ht, ss, cs) == 0, 'image.load: not a JPEG')
   local buf = ffi.new('unsigned char[?]', w[0] * ht[0] * 3)
   assert(lib.tjDecompress2(h, bytes, #bytes, buf, w[0], 0, ht[0], 0 --[[TJPF_RGB]], 0) == 0, 'image.load: JPEG decode failed')
   lib.tjDestroy(h)
   return from_bytes(buf, 3, ht[0], w[0])
end

function image.load(path, depth, _tensortype)
   local f = assert(io.open(path, 'rb'), 'image.load: cannot open ' .. tostring(path))
   local bytes = f:read('*a'); f:close()
   local img
   if bytes:sub(1, 2) == '\255\216' then img = load_jpeg(bytes)
   elseif bytes:sub(1, 1) == 'P' then img = load_pnm(bytes)
   else error('image.load: ' .. path .. ': only JPEG (libturbojpeg) and binary PPM / PGM are decoded here') end
   local C, H, W = dims(img)
   if depth == 3 and C == 1 then       -- grey file, three planes asked for
      local o = Host.new(3, H, W)
      for c = 1, 3 do o[c] = img[1] end
      img = o
   elseif depth == 1 and C == 3 then
      img = image.rgb2y(img)
   end
   return img
end

-- ------------------------------------------------------------------------------------------------ scale
-- one axis; src / dst are float pointers with element strides (the arithmetic order is dataset.py:_scale_axis's, fp32 step by step)
local f32 = ffi.typeof('float')
local function r(x) return tonumber(f32(x)) end      -- round to fp32 like every intermediate of the fp32 loader
local function scale_axis(src, ss, Ls, dst, ds, Ld)
   if Ld == Ls then for i = 0, Ls - 1 do dst[i * ds] = src[i * ss] end; return end
   if Ld > Ls then
      local scale = Ld > 1 and r((Ls - 1) / (Ld - 1)) or 0
      for d = 0, Ld - 2 do
         local sf = r(d * scale)
         local si = math.floor(sf)
         sf = r(sf - si)
         if Ls == 1 then dst[d * ds] = src[0]
         else dst[d * ds] = r(r(r(1 - sf) * src[si * ss]) + r(sf * src[(si + 1) * ss])) end
      end
      dst[(Ld - 1) * ds] = src[(Ls - 1) * ss]
      return
   end
   local scale = r(Ls / Ld)
   local i0, f0 = 0, 0
   for d = 0, Ld - 1 do
      local f1 = r((d + 1) * scale)
      local i1 = math.floor(f1)
      f1 = r(f1 - i1)
      local acc = r(r(1 - f0) * src[i0 * ss])
      local n = r(1 - f0)
      for si = i0 + 1, i1 - 1 do acc = r(acc + src[si * ss]); n = r(n + 1) end
      if i1 < Ls then acc = r(acc + r(f1 * src[i1 * ss])); n = r(n + f1) end
      dst[d * ds] = r(acc / n)
      i0, f0 = i1, f1
   end
end

function image.scale(img, w, h)
   local C, H, W = dims(img)
   local tmp, out = Host.new(C, H, w), Host.new(C, h, w)
   local s, t, o = img:data(), tmp:data(), out:data()
   for c = 0, C - 1 do
      for y = 0, H - 1 do scale_axis(s + (c * H + y) * W, 1, W, t + (c * H + y) * w, 1, w) end   -- rows to the target width
      for x = 0, w - 1 do scale_axis(t + c * H * w + x, w, H, o + c * h * w + x, w, h) end        -- columns to the target height
   end
   if #img:size() == 2 then return out:view(h, w) end
   return out
end

-- ------------------------------------------------------------------------------------------------ colour spaces
local function per_pixel(img, fn)    -- [3,H,W] -> [3,H,W]
   local C, H, W = dims(img)
   assert(C == 3, 'image: colour conversion expects 3 planes')
   local out = Host.new(3, H, W)
   local s, d, n = img:data(), out:data(), H * W
   for i = 0, n - 1 do d[i], d[n + i], d[2 * n + i] = fn(s[i], s[n + i], s[2 * n + i]) end
   return out
end
function image.rgb2y(img)            -- the reference's own weights (utils/nn_utils.lua:253-277)
   local C, H, W = dims(img)
   local out = Host.new(1, H, W)
   local s, d, n = img:data(), out:data(), H * W
   for i = 0, n - 1 do d[i] = 0.21 * s[i] + 0.72 * s[n + i] + 0.07 * s[2 * n + i] end
   return out
end
function image.rgb2yuv(img)
   return per_pixel(img, function(r_, g, b)
      return 0.299 * r_ + 0.587 * g + 0.114 * b, -0.14713 * r_ - 0.28886 * g + 0.436 * b, 0.615 * r_ - 0.51499 * g - 0.10001 * b end)
end
function image.yuv2rgb(img)
   return per_pixel(img, function(y, u, v) return y + 1.13983 * v, y - 0.39465 * u - 0.58060 * v, y + 2.03211 * u end)
end
function image.rgb2hsl(img)
   return per_pixel(img, function(r_, g, b)
      local mx, mn = math.max(r_, g, b), math.min(r_, g, b)
      local l = (mx + mn) / 2
      if mx == mn then return 0, 0, l end
      local d = mx - mn
      local s = l > 0.5 and d / (2 - mx - mn) or d / (mx + mn)
      local h
      if mx == r_ then h = (g - b) / d + (g < b and 6 or 0) elseif mx == g then h = (b - r_) / d + 2 else h = (r_ - g) / d + 4 end
      return h / 6, s, l
   end)
end
local function hue(p, q, t)
   if t < 0 then t = t + 1 end
   if t > 1 then t = t - 1 end
   if t < 1 / 6 then return p + (q - p) * 6 * t end
   if t < 1 / 2 then return q end
   if t < 2 / 3 then return p + (q - p) * (2 / 3 - t) * 6 end
   return p
end
function image.hsl2rgb(img)
   return per_pixel(img, function(h, s, l)
      if s == 0 then return l, l, l end
      local q = l < 0.5 and l * (1 + s) or l + s - l * s
      local p = 2 * l - q
      return hue(p, q, h + 1 / 3), hue(p, q, h), hue(p, q, h - 1 / 3)
   end)
end

-- ------------------------------------------------------------------------------------------------ save
local crc_table = nil
local function crc32(s)
   if not crc_table then
      crc_table = {}
      for i = 0, 255 do
         local c = i
         for _ = 1, 8 do c = bit.band(c, 1) == 1 and bit.bxor(0xEDB88320, bit.rshift(c, 1)) or bit.rshift(c, 1) end
         crc_table[i] = c
      end
   end
   local c = 0xFFFFFFFF
   for i = 1, #s do c = bit.bxor(crc_table[bit.band(bit.bxor(c, s:byte(i)), 0xFF)], bit.rshift(c, 8)) end
   return bit.bxor(c, 0xFFFFFFFF)
end
local function be32(v)
   v = v % 4294967296
   return string.char(math.floor(v / 16777216) % 256, math.floor(v / 65536) % 256, math.floor(v / 256) % 256, v % 256)
end
local function adler32(s)
   local a, b = 1, 0
   for i = 1, #s do a = (a + s:byte(i)) % 65521; b = (b + a) % 65521 end
   return b * 65536 + a
end
local function zlib_stored(raw)     -- a valid zlib stream of uncompressed ("stored") deflate blocks
   local out, pos, n = { '\120\1' }, 1, #raw
   repeat
      local len = math.min(65535, n - pos + 1)
      local final = (pos + len > n) and 1 or 0
      out[#out + 1] = string.char(final, len % 256, math.floor(len / 256), (65535 - len) % 256, math.floor((65535 - len) / 256))
      out[#out + 1] = raw:sub(pos, pos + len - 1)
      pos = pos + len
   until pos > n
   out[#out + 1] = be32(adler32(raw))
   return table.concat(out)
end
local function chunk(tag, data) return be32(#data) .. tag .. data .. be32(crc32(tag .. data)) end

local function to_bytes(img)          -- FloatTensor [C,H,W] in [0,1] -> rows of interleaved 8-bit pixels
   local C, H, W = dims(img)
   assert(C == 1 or C == 3, 'image.save: 1 or 3 planes')
   local s = img:data()
   local rows = {}
   for y = 0, H - 1 do
      local px = {}
      for x = 0, W - 1 do
         for c = 0, C - 1 do
            local v = s[(c * H + y) * W + x]
            px[#px + 1] = string.char(math.max(0, math.min(255, math.floor(v * 255 + 0.5))))
         end
      end
      rows[y + 1] = table.concat(px)
   end
   return rows, C, H, W
end
function image.save(path, img)
   if img.__typename ~= 'torch.FloatTensor' then img = img:float() end
   local rows, C, H, W = to_bytes(img)
   local f = assert(io.open(path, 'wb'), 'image.save: cannot write ' .. tostring(path))
   if path:lower():match('%.p[pgn]m$') then
      f:write(string.format('%s\n%d %d\n255\n', C == 3 and 'P6' or 'P5', W, H), table.concat(rows))
   else
      local raw = {}
      for y = 1, H do raw[y] = '\0' .. rows[y] end     -- filter type 0 in front of every scanline
      local ihdr = be32(W) .. be32(H) .. string.char(8, C == 3 and 2 or 0, 0, 0, 0)
      f:write('\137PNG\r\n\26\n', chunk('IHDR', ihdr), chunk('IDAT', zlib_stored(table.concat(raw))), chunk('IEND', ''))
   end
   f:close()
end

-- ------------------------------------------------------------------------------------------------ grids
function image.toDisplayTensor(opts)
   local input, pad = opts.input or opts[1], opts.padding or 0
   local list = {}
   if type(input) == 'table' and not input.__tensor then list = input
   else for i = 1, input:size(1) do list[i] = input[i] end end
   local n = #list
   local C, H, W = dims(list[1])
   local nrow = opts.nrow or math.ceil(math.sqrt(n))
   local ncol = math.ceil(n / nrow)
   local out = Host.new(C, ncol * (H + pad) + pad, nrow * (W + pad) + pad):zero()
   local o, OW, OH = out:data(), nrow * (W + pad) + pad, ncol * (H + pad) + pad
   for i = 1, n do
      local gy, gx = math.floor((i - 1) / nrow), (i - 1) % nrow
      local s = list[i]:data()
      for c = 0, C - 1 do
         for y = 0, H - 1 do
            local dst = (c * OH + pad + gy * (H + pad) + y) * OW + pad + gx * (W + pad)
            for x = 0, W - 1 do o[dst + x] = s[(c * H + y) * W + x] end
         end
      end
   end
   return out
end
function image.display(opts)          -- adversarial.lua:346 (visualizeNetwork, qlua only): a file instead of a window
   local input = opts.image or opts[1]
   local grid = (type(input) == 'table' and not input.__tensor) and image.toDisplayTensor({ input = input, padding = opts.padding or 1, nrow = opts.nrow })
                or input
   local dir = ((_G.OPT and _G.OPT.save) or 'logs') .. '/display'
   os.execute(string.format('mkdir -p %q', dir))
   image.save(string.format('%s/%s.png', dir, tostring(opts.legend or opts.win or 'image'):gsub('[^%w_-]', '_')), grid)
   return opts.win
end

_G.image = image
return image
