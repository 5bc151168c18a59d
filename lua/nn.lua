-- shim: `require 'nn'` (train.lua:101, models.lua:2) -> the engine's module classes
nn = require('catgan').nn
return nn
