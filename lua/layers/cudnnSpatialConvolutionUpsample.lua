-- shim: `require 'layers.cudnnSpatialConvolutionUpsample'` (models.lua:5): cudnn.SpatialConvolutionUpsample is an engine class
require 'cudnn'
return cudnn.SpatialConvolutionUpsample
