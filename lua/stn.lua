-- shim: `require 'stn'` (models.lua:6): nn.AffineTransformMatrixGenerator / AffineGridGeneratorBHWD / BilinearSamplerBHWD live in nn
require 'nn'
return nn
