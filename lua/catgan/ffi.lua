--[[ catgan.ffi — LuaJIT FFI binding of include/catgan.h: the header itself is fed to ffi.cdef (comments, preprocessor
lines and the extern "C" guard stripped), so this binding, the ctypes one (cat-generator_amd/_abi.py) and the library
cannot drift apart.  Every int-returning entry point is a status code: check() raises cg_last_error(). ]]
local ffi = require 'ffi'

local function load_header(path)
   local f = assert(io.open(path), 'cannot open ' .. path .. ' (set CATGAN_HEADER)')
   local src = f:read('*a'); f:close()
   src = src:gsub('/%*.-%*/', ''):gsub('//[^\n]*', '')
   src = src:gsub('\n%s*#[^\n]*', '\n')                                   -- #ifndef / #define / #include / #endif
   src = src:gsub('extern%s+"C"%s*{', ''):gsub('\n}%s*\n', '\n')            -- the C++ guard
   ffi.cdef(src)
end
load_header(os.getenv('CATGAN_HEADER') or 'include/catgan.h')
local C = ffi.load(os.getenv('CATGAN_LIB') or 'cat-generator_amd/lib/libcatgan_hip.so')
assert(C.cg_abi_version() == 1, 'libcatgan_hip.so: ABI version mismatch')

local function check(rc)
   if rc ~= 0 then error(ffi.string(C.cg_last_error()), 2) end
end

return { C = C, check = check, ffi = ffi }
