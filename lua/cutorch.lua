-- shim: `require 'cutorch'` (train.lua:102,109-110): device selection and seeding
local cg = require 'catgan'
cutorch = { setDevice = function(i) cg.setDevice(i - 1) end, manualSeed = function(s) cg.manualSeed(s) end,
            synchronize = cg.synchronize, getDeviceCount = function() local n = require('ffi').new('int[1]'); cg.check(cg.C.cg_device_count(n)); return n[0] end }
return cutorch
