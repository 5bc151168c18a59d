-- shim: `require 'layers.SpatialConvolutionUpsample'`: nn.SpatialConvolutionUpsample is an engine class
require 'nn'
return nn.SpatialConvolutionUpsample
