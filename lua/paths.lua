--[[ shim: `require 'paths'` (train.lua:4, dataset.lua:3): paths.concat (train.lua:119,253), paths.filep (train.lua:255),
paths.dirp, paths.files (dataset.lua:65: an iterator over a directory's entries), paths.basename / dirname / mkdir.
POSIX only (the engine runs on Linux); directory listing through `ls -a` (no lfs in a bare LuaJIT). ]]
local paths = {}

function paths.concat(...)
   local parts = { ... }
   local out = nil
   for _, p in ipairs(parts) do
      p = tostring(p)
      if out == nil or p:sub(1, 1) == '/' then out = p
      elseif out:sub(-1) == '/' then out = out .. p
      else out = out .. '/' .. p end
   end
   return out or ''
end
function paths.filep(p)
   local f = io.open(p, 'rb')
   if not f then return false end
   local ok = f:read(0) ~= nil or f:seek('end') == 0     -- a directory opens but cannot be read
   f:close()
   return ok and not paths.dirp(p)
end
function paths.dirp(p)
   local f = io.open(p .. '/.', 'rb')
   if f then f:close(); return true end
   return false
end
function paths.basename(p, ext)
   local b = p:match('([^/]+)/*$') or p
   if ext and b:sub(-#ext) == ext then b = b:sub(1, -#ext - 1) end
   return b
end
function paths.dirname(p)
   local d = p:match('^(.*)/[^/]*$')
   if d == nil then return '.' end
   if d == '' then return '/' end
   return d
end
function paths.mkdir(p) return os.execute(string.format('mkdir -p %q', p)) end
function paths.files(dir)     -- iterator: every entry of the directory, '.' and '..' included (as torch's)
   local h = io.popen(string.format('ls -a %q 2>/dev/null', dir))
   local names = {}
   if h then for l in h:lines() do names[#names + 1] = l end; h:close() end
   local i = 0
   return function() i = i + 1; return names[i] end
end

_G.paths = paths
return paths
