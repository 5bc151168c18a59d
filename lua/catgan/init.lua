--[[ catgan — the engine's LuaJIT front: torch-lite tensors, nn / cudnn / stn classes, optim, collectives.

    package.path = 'lua/?.lua;lua/?/init.lua;' .. package.path    -- BEFORE the reference's own requires
    -- now `require 'torch'`, `require 'nn'`, `require 'cunn'`, `require 'cudnn'`, `require 'stn'`, `require 'optim'`,
    -- `require 'LeakyReLU'` ... resolve to the shims in lua/ (each a few lines) and models.lua / adversarial.lua /
    -- weight-init.lua / utils/nn_utils.lua run as written.

NOT EXECUTED in the build container (no Lua of any kind there; SURVEY.md Appendix C).  What stands in for that:
  * scripts/check_lua_binding.py: every C.cg_* call in lua/ against include/catgan.h (name and arity), block structure of every file,
    the builder tables of catgan.net against planned.py / csrc/net.hip, and - round 4 - every `require`, every `namespace.function`
    call and every method name of the reference's adversarial.lua / models.lua / train.lua / weight-init.lua / utils/nn_utils.lua /
    dataset.lua resolved against a provider in lua/ (tests/golden/lua_reference_names.json holds the list for machines without
    /root/reference);
  * cat-generator_amd/nn.py with nn.planned = False is the executable twin of catgan.nn, class for class and call for call;
    cat-generator_amd/t7.py that of catgan.t7 (torch.save / torch.load);
  * tools/abi_replay.cpp replays the call sequence of a whole G+D update - the cg_net_* sequence catgan.net issues at the benchmarked
    batch - through the same entry points with no interpreter and no PyTorch in the process, bit-equal to the Python host
    (tests/test_abi_step.py). ]]
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
