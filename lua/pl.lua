--[[ shim: `require 'pl'` (adversarial.lua:3, train.lua:3).  The reference pulls Penlight in for ONE thing: the global `lapp`
that parses train.lua:14-49's option text.  This is that function and nothing else of Penlight: a spec line is
    --name   (default VALUE)   description        -> OPT.name = VALUE (number if it reads as one, else the quoted string)
    --flag                      description        -> OPT.flag = false unless given
    -s,--name ...                                  -> short alias
and the command line (`arg`) overrides: `--name value`, `--name=value`, `--flag`.  Unknown options are an error, as in lapp. ]]
local function convert(s)
   local q = s:match('^"(.*)"$') or s:match("^'(.*)'$")
   if q then return q end
   return tonumber(s) or s
end

local lapp = {}
function lapp.parse(spec, args)
   args = args or _G.arg or {}
   local opt, kind, alias = {}, {}, {}
   for line in spec:gmatch('[^\n]+') do
      local short, name = line:match('^%s*%-(%w),%-%-([%w_]+)')
      if not name then name = line:match('^%s*%-%-([%w_]+)') end
      if name then
         local def = line:match('%(default%s+(.-)%)')
         if def then opt[name], kind[name] = convert(def), 'value' else opt[name], kind[name] = false, 'flag' end
         if short then alias[short] = name end
      end
   end
   local i = 1
   while i <= #args do
      local a = args[i]
      local name, val = a:match('^%-%-([%w_]+)=(.*)$')
      if not name then name = a:match('^%-%-([%w_]+)$') end
      if not name then local s = a:match('^%-(%w)$'); name = s and alias[s] end
      if not name or kind[name] == nil then error('lapp: unknown option ' .. tostring(a)) end
      if kind[name] == 'flag' then
         opt[name] = true
      else
         if val == nil then i = i + 1; val = args[i] end
         if val == nil then error('lapp: option --' .. name .. ' needs a value') end
         local v = convert(val)
         if type(opt[name]) == 'number' and type(v) ~= 'number' then error('lapp: option --' .. name .. ' expects a number') end
         opt[name] = v
      end
      i = i + 1
   end
   return opt
end
setmetatable(lapp, { __call = function(_, spec, args) return lapp.parse(spec, args) end })

_G.lapp = lapp
return { lapp = lapp }
