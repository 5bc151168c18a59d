--[[ shim: `pcall(require, 'display')` (train.lua:5): the browser display server of szym/display.  The engine has none;
DISP.image / DISP.plot (utils/nn_utils.lua:168-182) write what they are given to files under OPT.save instead - the grids as PNG
through image.save, the plot data as tab-separated text - so `th train.lua` without --noplot keeps running. ]]
local display = { dir = nil }
local function outdir()
   local d = display.dir or ((_G.OPT and _G.OPT.save) or 'logs') .. '/display'
   os.execute(string.format('mkdir -p %q', d))
   return d
end
function display.image(img, opts)
   opts = opts or {}
   local grid = img
   if type(img) == 'table' and not img.__tensor then     -- a list of images: one grid
      grid = require('image').toDisplayTensor({ input = img, padding = 1 })
   end
   local fn = string.format('%s/win%s.png', outdir(), tostring(opts.win or 0))
   require('image').save(fn, grid)
   return opts.win
end
function display.plot(data, opts)
   opts = opts or {}
   local f = assert(io.open(string.format('%s/plot%s.tsv', outdir(), tostring(opts.win or 0)), 'w'))
   if opts.labels then f:write(table.concat(opts.labels, '\t'), '\n') end
   for _, row in ipairs(data) do
      local cells = {}
      for i, v in ipairs(row) do cells[i] = tostring(v) end
      f:write(table.concat(cells, '\t'), '\n')
   end
   f:close()
   return opts.win
end
return display
