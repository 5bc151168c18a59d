--[[ shim: the `sys` package the `th` launcher preloads: sys.clock (adversarial.lua:34,278), sys.dirname (train.lua:254,
utils/nn_utils.lua:581). ]]
local ffi = require 'ffi'
ffi.cdef [[ typedef struct { long tv_sec; long tv_usec; } cg_timeval; int gettimeofday(cg_timeval*, void*); ]]
local sys = {}
function sys.clock()     -- wall-clock seconds with sub-second resolution (torch's sys.clock is gettimeofday too)
   local tv = ffi.new('cg_timeval')
   ffi.C.gettimeofday(tv, nil)
   return tonumber(tv.tv_sec) + tonumber(tv.tv_usec) * 1e-6
end
function sys.dirname(p) return require('paths').dirname(p) end
function sys.basename(p) return require('paths').basename(p) end
function sys.execute(cmd) local h = io.popen(cmd); local s = h:read('*a'); h:close(); return (s:gsub('%s+$', '')) end
_G.sys = sys
return sys
