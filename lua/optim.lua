-- shim: `require 'optim'` -> optim.adam / sgd / adagrad / ConfusionMatrix on the engine
optim = require('catgan').optim
return optim
