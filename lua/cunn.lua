-- shim: `require 'cunn'` (train.lua:103): the device implementations ARE the classes of `nn` here
require 'nn'
return nn
