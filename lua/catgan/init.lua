--[[ catgan — the engine's LuaJIT front: torch-lite tensors, nn / cudnn / stn classes, optim, collectives.

    package.path = 'lua/?.lua;lua/?/init.lua;' .. package.path    -- BEFORE the reference's own requires
    -- now `require 'torch'`, `require 'nn'`, `require 'cunn'`, `require 'cudnn'`, `require 'stn'`, `require 'optim'`,
    -- `require 'LeakyReLU'` ... resolve to the shims in lua/ (each a few lines) and models.lua / adversarial.lua /
    -- weight-init.lua / utils/nn_utils.lua run as written.

NOT EXECUTED in the build container (no Lua of any kind there; SURVEY.md Appendix C).  Three things stand in for that:
scripts/check_lua_binding.py verifies every C.cg_* call in these files against include/catgan.h (name and arity);
cat-generator_amd/nn.py with nn.fusion = False is the executable twin, class for class and call for call; and
tools/abi_step.cpp drives a whole G+D update through the same entry points with no interpreter and no PyTorch in the
process (tests/test_abi_step.py compares it with the Python host). ]]
local abi = require 'catgan.ffi'
local T = require 'catgan.tensor'
local N = require 'catgan.nn'
local M = { C = abi.C, check = abi.check, torch = T.torch, Tensor = T.Device, FloatTensor = T.Host, nn = N.nn, cudnn = N.cudnn,
            optim = require 'catgan.optim' }

function M.manualSeed(seed) N.nn.rng.seed, N.nn.rng.offset = seed, 0; math.randomseed(seed) end
function M.set_stream(s) T.stream = s end
function M.setDevice(i) abi.check(abi.C.cg_set_device(i)) end   -- cutorch.setDevice(OPT.gpu + 1) is 1-based: pass OPT.gpu
function M.synchronize() abi.check(abi.C.cg_stream_sync(T.stream)) end
function M.comm() return require 'catgan.comm' end
return M
