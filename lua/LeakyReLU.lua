-- shim: `require 'LeakyReLU'` (models.lua:3).  The reference's in-tree file defines nn.LeakyReLU with tensor arithmetic
-- (LeakyReLU.lua:13-31); the engine's class (same constructor, same semantics incl. x == 0) is already in nn.
require 'nn'
return nn.LeakyReLU
