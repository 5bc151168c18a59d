-- shim: `require 'cudnn'` (models.lua:206 uses cudnn.SpatialConvolution; layers/cudnnSpatialConvolutionUpsample.lua)
cudnn = require('catgan').cudnn
return cudnn
