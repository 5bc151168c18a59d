-- shim: `require 'torch'` (train.lua:1, models.lua:1) -> the engine's torch-lite, plus what the `th` launcher puts in the global
-- environment next to it (paths, sys, xlua: train.lua / adversarial.lua use them without requiring them)
torch = require('catgan').torch
require 'catgan.torchx'
paths = require 'paths'
sys = require 'sys'
xlua = require 'xlua'
return torch
