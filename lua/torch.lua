-- shim: `require 'torch'` (train.lua:1, models.lua:1) -> the engine's torch-lite
torch = require('catgan').torch
return torch
