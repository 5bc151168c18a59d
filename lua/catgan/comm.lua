--[[ catgan.comm — the data-parallel exchanges of the step through cg_comm_* (csrc/comm.hip: RCCL over xGMI on a side
HIP stream with event fork / join).  One LuaJIT process per GPU (SURVEY.md 8e); the reference itself is single-GPU
(train.lua:108-112).

   local comm = require 'catgan.comm'
   comm.init(rank, nranks, '/shared/path/uid')      -- rank 0 writes the two RCCL unique ids, the others read them; the launcher's
                                                    -- CATGAN_RUN_ID (any string unique to this launch) becomes part of the file name, so
                                                    -- a file a previous run left behind is never read
   -- in fevalD / fevalG_on_D, right after MODEL_X:backward and BEFORE the penalty / clamp lines (adversarial.lua:89-112):
   comm.allreduce_mean(GRAD_PARAMETERS_D)           -- or comm.allreduce_mean_async(...) ... comm.wait()
   -- or, with the planned executor, in buckets that travel under the backward itself (both nets since round 6: Linear(20480, 256), 79 % of
   -- D's gradient, is the first bucket D's backward completes): set MODEL_X._bucket_overlap = true before the first pass, then
   --    MODEL_X:backward(input, gradOutput)
   --    if MODEL_X:finishBuckets(comm) == 0 then comm.allreduce_mean(GRAD_PARAMETERS_X) end   -- 0: no contiguous buckets
installs nn.sync_bn so that nn.SpatialBatchNormalization all-reduces its fp64 sums (sync-BN). ]]
local ffi = require 'ffi'
local abi = require 'catgan.ffi'
local T = require 'catgan.tensor'
local C, check = abi.C, abi.check

local comm = { grad = nil, bn = nil, nranks = 1, rank = 0 }

local function read_file(path, n)
   for _ = 1, 6000 do   -- wait up to ~60 s for rank 0
      local f = io.open(path, 'rb')
      if f then local s = f:read('*a'); f:close(); if #s == n then return s end end
      ffi.C.usleep(10000)
   end
   error('catgan.comm: timed out waiting for ' .. path)
end

function comm.init(rank, nranks, uid_path, device)
   ffi.cdef 'int usleep(unsigned int);'
   check(C.cg_set_device(device or rank))
   local ok = ffi.new('int[1]'); check(C.cg_comm_available(ok))
   assert(ok[0] == 1, 'librccl.so.1 not loadable')
   local ids
   uid_path = uid_path .. '.' .. (os.getenv('CATGAN_RUN_ID') or os.getenv('MASTER_PORT') or '0')
   if rank == 0 then
      os.remove(uid_path)                                 -- never let a reader see the previous launch's ids
      local a, b = ffi.new('char[128]'), ffi.new('char[128]')
      check(C.cg_comm_unique_id(a, 128)); check(C.cg_comm_unique_id(b, 128))
      ids = ffi.string(a, 128) .. ffi.string(b, 128)
      local f = assert(io.open(uid_path .. '.tmp', 'wb')); f:write(ids); f:close()
      os.rename(uid_path .. '.tmp', uid_path)
   else
      ids = read_file(uid_path, 256)
   end
   local function mk(off)
      local h = ffi.new('void*[1]')
      check(C.cg_comm_init(h, nranks, rank, ids:sub(off + 1, off + 128), 128))
      return ffi.gc(h[0], function(p) C.cg_comm_destroy(p) end)
   end
   comm.grad, comm.bn, comm.nranks, comm.rank = mk(0), mk(128), nranks, rank
   if rank == 0 then os.remove(uid_path) end              -- ncclCommInitRank is collective: every rank has read the ids by now
   -- sync-BN hook of catgan.nn: all-reduce(sum) of `count` doubles, returns the global element count
   require('catgan.nn').nn.sync_bn = function(sums, count, M)
      check(C.cg_comm_allreduce(comm.bn, T.stream, sums, count, 1, 0)); check(C.cg_comm_wait(comm.bn, T.stream))
      return M * nranks
   end
end

function comm.allreduce_mean_async(t) check(C.cg_comm_allreduce(comm.grad, T.stream, t.ptr, t.n, 0, 1)) end
function comm.wait() check(C.cg_comm_wait(comm.grad, T.stream)) end
function comm.allreduce_mean(t) comm.allreduce_mean_async(t); comm.wait() end
function comm.broadcast(t, root) check(C.cg_comm_broadcast(comm.grad, T.stream, t.ptr, t.n, 0, root or 0)); comm.wait() end

return comm
