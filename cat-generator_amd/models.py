"""models.lua's generator / discriminator factories, translated line for line onto the engine's nn layer.

Only the definitions on the hot path are here: G32up-c (models.lua:196-228, the default
via create_G :234-240), G32up (:138-160), D32_st3 (:640-711, the only D create_D returns, :276) and the
spatial-transformer factory (:814-906).  `create_G32up_c_64` is the builder-defined 64x64 extension of
BASELINE.json config #5.
"""
from . import cudnn, nn
from .tensor import Tensor
from .weight_init import w_init


def create_G_decoder_upsampling32(dimensions, noiseDim):
    """models.lua:138-160."""
    model = nn.Sequential()
    model.add(nn.Linear(noiseDim, 128 * 8 * 8))
    model.add(nn.View(128, 8, 8))
    model.add(nn.PReLU(None, None, True))

    model.add(nn.SpatialUpSamplingNearest(2))
    model.add(cudnn.SpatialConvolution(128, 256, 5, 5, 1, 1, (5 - 1) // 2, (5 - 1) // 2))
    model.add(nn.SpatialBatchNormalization(256))
    model.add(nn.PReLU(None, None, True))

    model.add(nn.SpatialUpSamplingNearest(2))
    model.add(cudnn.SpatialConvolution(256, 128, 5, 5, 1, 1, (5 - 1) // 2, (5 - 1) // 2))
    model.add(nn.SpatialBatchNormalization(128))
    model.add(nn.PReLU(None, None, True))

    model.add(cudnn.SpatialConvolution(128, dimensions[0], 3, 3, 1, 1, (3 - 1) // 2, (3 - 1) // 2))
    model.add(nn.Sigmoid())

    model = w_init(model, "heuristic")
    return model


def create_G_decoder_upsampling32c(dimensions, noiseDim, base=4):
    """models.lua:196-228.  `base` is 4 in the reference (4x4 -> 32x32); base=8 gives the 64x64 extension."""
    model = nn.Sequential()
    # 4x4
    model.add(nn.Linear(noiseDim, 512 * base * base))
    model.add(nn.PReLU(None, None, True))
    model.add(nn.View(512, base, base))

    # 4x4 -> 8x8
    model.add(nn.SpatialUpSamplingNearest(2))
    model.add(cudnn.SpatialConvolution(512, 512, 3, 3, 1, 1, (3 - 1) // 2, (3 - 1) // 2))
    model.add(nn.SpatialBatchNormalization(512))
    model.add(nn.PReLU(None, None, True))

    # 8x8 -> 16x16
    model.add(nn.SpatialUpSamplingNearest(2))
    model.add(cudnn.SpatialConvolution(512, 256, 3, 3, 1, 1, (3 - 1) // 2, (3 - 1) // 2))
    model.add(nn.SpatialBatchNormalization(256))
    model.add(nn.PReLU(None, None, True))

    # 16x16 -> 32x32
    model.add(nn.SpatialUpSamplingNearest(2))
    model.add(cudnn.SpatialConvolution(256, 128, 5, 5, 1, 1, (5 - 1) // 2, (5 - 1) // 2))
    model.add(nn.SpatialBatchNormalization(128))
    model.add(nn.PReLU(None, None, True))

    model.add(cudnn.SpatialConvolution(128, dimensions[0], 3, 3, 1, 1, (3 - 1) // 2, (3 - 1) // 2))
    model.add(nn.Sigmoid())

    model = w_init(model, "heuristic")
    return model


def create_G32up_c_64(dimensions, noiseDim):
    """BASELINE.json config #5: G32up-c scaled to 64x64 (first Linear -> 512*8*8)."""
    return create_G_decoder_upsampling32c(dimensions, noiseDim, base=8)


def create_G(dimensions, noiseDim):
    """models.lua:234-240 (the 16px generator is an unused variant, out of scope)."""
    if dimensions[1] == 16:
        raise NotImplementedError("create_G_decoder_upsampling16 is not on the hot path (SURVEY.md §2.1 row 2)")
    if dimensions[1] == 64:
        return create_G32up_c_64(dimensions, noiseDim)
    return create_G_decoder_upsampling32c(dimensions, noiseDim)


def create_D(dimensions, cuda=False):
    """models.lua:268-277: always st3."""
    return create_D32_st3(dimensions, cuda)


def create_D32_st3(dimensions, cuda=False):
    """models.lua:640-711.  `cuda` only decides whether the nn.Copy host<->device layers wrap the net
    (:642-644, :703-706); the arithmetic always runs on the device."""
    conv = nn.Sequential()
    if cuda:
        conv.add(nn.Copy("torch.FloatTensor", "torch.CudaTensor", True, True))
    conv.add(createSpatialTransformer(True, False, False, dimensions[1], dimensions[0], cuda))
    conv.add(nn.SpatialConvolution(dimensions[0], 64, 3, 3, 1, 1, (3 - 1) // 2))
    conv.add(nn.PReLU(None, None, True))
    conv.add(nn.SpatialConvolution(64, 64, 3, 3, 1, 1, (3 - 1) // 2))
    conv.add(nn.PReLU(None, None, True))
    conv.add(nn.SpatialAveragePooling(2, 2, 2, 2))
    conv.add(nn.SpatialDropout(0.2))

    def st_branch():
        b = nn.Sequential()
        b.add(createSpatialTransformer(True, True, True, dimensions[1] // 2, 64, cuda))
        b.add(nn.SpatialConvolution(64, 64, 3, 3, 1, 1, (3 - 1) // 2))
        b.add(nn.PReLU(None, None, True))
        b.add(nn.SpatialMaxPooling(2, 2))
        b.add(nn.SpatialDropout(0.2))
        b.add(nn.SpatialConvolution(64, 64, 3, 3, 1, 1, (3 - 1) // 2))
        b.add(nn.PReLU(None, None, True))
        return b

    branch1 = st_branch()
    branch2 = st_branch()
    branch3 = st_branch()

    branch4 = nn.Sequential()
    branch4.add(nn.SpatialConvolution(64, 128, 5, 5, 1, 1, (5 - 1) // 2))
    branch4.add(nn.PReLU(None, None, True))
    branch4.add(nn.SpatialMaxPooling(2, 2))
    branch4.add(nn.SpatialDropout(0.2))
    branch4.add(nn.SpatialConvolution(128, 128, 7, 7, 1, 1, (7 - 1) // 2))
    branch4.add(nn.PReLU(None, None, True))

    concy = nn.Concat(2)
    concy.add(branch1)
    concy.add(branch2)
    concy.add(branch3)
    concy.add(branch4)

    conv.add(concy)
    conv.add(nn.SpatialDropout())
    feat = (64 + 64 + 64 + 128) * (dimensions[1] // 4) * (dimensions[2] // 4)
    conv.add(nn.View(feat))
    conv.add(nn.Linear(feat, 256))
    conv.add(nn.PReLU(None, None, True))
    conv.add(nn.Dropout())
    conv.add(nn.Linear(256, 1))
    conv.add(nn.Sigmoid())

    if cuda:
        conv.add(nn.Copy("torch.CudaTensor", "torch.FloatTensor", True, True))
        conv.cuda()

    conv = w_init(conv, "heuristic")
    return conv


def createSpatialTransformer(allow_rotation, allow_scaling, allow_translation, input_size, input_channels, cuda=True):
    """models.lua:814-906.  The reference forces the sampler onto the CPU in GPU mode (:889-899) and wraps it in
    nn.Copy layers; the engine's sampler is a device kernel, so those copies (pure transport) are not added."""
    init_bias = []
    nbr_params = 0
    if allow_rotation:
        nbr_params += 1
        init_bias.append(0)
    if allow_scaling:
        nbr_params += 1
        init_bias.append(1)
    if allow_translation:
        nbr_params += 2
        init_bias += [0, 0]
    if nbr_params == 0:
        raise NotImplementedError("fully parametrised transformer (6 params) is not used by create_D32_st3")

    # localization network
    net = nn.Sequential()
    net.add(nn.SpatialAveragePooling(2, 2, 2, 2))
    net.add(nn.SpatialConvolution(input_channels, 16, 3, 3, 1, 1, (3 - 1) // 2))
    net.add(nn.LeakyReLU())
    net.add(nn.SpatialConvolution(16, 16, 3, 3, 1, 1, (3 - 1) // 2))
    net.add(nn.LeakyReLU())
    net.add(nn.SpatialAveragePooling(2, 2, 2, 2))

    newHeight = input_size // 4
    net.add(nn.View(16 * newHeight * newHeight))
    net.add(nn.Linear(16 * newHeight * newHeight, 64))
    net.add(nn.LeakyReLU())
    classifier = nn.Linear(64, nbr_params)
    net.add(classifier)

    net = w_init(net, "heuristic")
    # identity initialisation (see paper, A.3 section): models.lua:859-860
    classifier.weight.zero()
    classifier.bias = Tensor.from_numpy(init_bias)  # replaces the tensor object, as the reference does

    localization_network = net

    ct = nn.ConcatTable()
    branch1 = nn.Sequential()
    branch1.add(nn.Transpose((3, 4), (2, 4)))
    branch2 = nn.Sequential()
    branch2.add(localization_network)
    branch2.add(nn.AffineTransformMatrixGenerator(allow_rotation, allow_scaling, allow_translation))
    branch2.add(nn.AffineGridGeneratorBHWD(input_size, input_size))
    ct.add(branch1)
    ct.add(branch2)

    st = nn.Sequential()
    st.add(ct)
    sampler = nn.BilinearSamplerBHWD()
    st.add(sampler)
    st.add(nn.Transpose((2, 4), (3, 4)))
    return st
