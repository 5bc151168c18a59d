"""`cudnn.SpatialConvolution(nIn,nOut,kW,kH,dW,dH,padW,padH)` alias (models.lua:206,212,218,222).

Same kernel as nn.SpatialConvolution; the distinct __typename matters because weight-init.lua:54 tests for
'nn.SpatialConvolution' only, so G's convolutions keep the default reset() but get their bias zeroed."""
from . import nn


class SpatialConvolution(nn.SpatialConvolution):
    _typename = "cudnn.SpatialConvolution"
