--[[ shim: `xlua` (preloaded by `th`): xlua.progress(current, total) - the one call inside the training loop
(adversarial.lua:270).  A plain carriage-return progress line at most ten times a second: nothing here may cost the loop
time, the step it sits in takes 6.5 ms. ]]
local xlua = {}
local last = 0
function xlua.progress(cur, total)
   local now = os.clock()
   if cur < total and now - last < 0.1 then return end
   last = now
   local width = 40
   local done = math.floor(width * math.min(1, cur / math.max(total, 1)))
   io.write(string.format('\r [%s%s] %d/%d', string.rep('=', done), string.rep('.', width - done), cur, total))
   if cur >= total then io.write('\n') end
   io.flush()
end
function xlua.print(...) print(...) end
_G.xlua = xlua
return xlua
