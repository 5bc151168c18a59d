"""weight-init.lua restated (weight-init.lua:14-75): `require('weight-init')(net, 'heuristic')`."""
import math


def w_init_heuristic(fan_in, fan_out):   # weight-init.lua:14-16
    return math.sqrt(1.0 / (3.0 * fan_in))


def w_init_xavier(fan_in, fan_out):      # :21-23
    return math.sqrt(2.0 / (fan_in + fan_out))


def w_init_xavier_caffe(fan_in, fan_out):  # :28-30
    return math.sqrt(1.0 / fan_in)


def w_init_kaiming(fan_in, fan_out):     # :35-37
    return math.sqrt(4.0 / (fan_in + fan_out))


_METHODS = {"heuristic": w_init_heuristic, "xavier": w_init_xavier, "xavier_caffe": w_init_xavier_caffe,
            "kaiming": w_init_kaiming}


def w_init(net, arg):
    """Loops over the TOP-LEVEL modules only (weight-init.lua:52), exactly like the reference: nested
    containers (nn.Concat branches, spatial transformers) are not visited."""
    assert arg in _METHODS  # weight-init.lua:48
    method = _METHODS[arg]
    for m in net.modules:
        tn = m.typename
        if tn in ("nn.SpatialConvolution", "nn.SpatialConvolutionMM"):
            m.reset(method(m.nInputPlane * m.kH * m.kW, m.nOutputPlane * m.kH * m.kW))
        elif tn == "nn.Linear":
            m.reset(method(m.weight.size(2), m.weight.size(1)))
        if getattr(m, "bias", None) is not None:
            m.bias.zero()
    return net
