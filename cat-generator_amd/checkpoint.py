"""Checkpoint / resume of a training state (train.lua:127-142, 241-244, 252-261).

The reference saves {D, G, opt, plot_data, epoch} with torch.save after NN_UTILS.prepareNetworkForSave, and on
resume reads an `optstate` field it never wrote (train.lua:132 vs :260) — so Adam's moments silently restart.
Here one .npz holds what is needed to continue bit-for-bit: the two flat parameter vectors (Torch7
getParameters() order, canonical layouts), the optimiser state (t, m, v / variances / momentum buffers), the
batch-norm running statistics, EPOCH, OPT, the position of the counter-based RNG stream (host offset AND the device-side
base a hipGraph replay advances), the host generator that draws the real-batch indices (math.random, adversarial.lua:226)
and the dataset shuffle generator (torch.randperm, dataset.lua:158)."""
import json

import numpy as np
import torch

from . import dataset, nn
from .tensor import Tensor, rng


def _pack_rs(prefix, rs, out):
    kind, keys, pos, has_gauss, cached = rs.get_state()
    out[prefix + "_keys"] = np.asarray(keys, dtype=np.uint32)
    out[prefix + "_meta"] = np.array([pos, has_gauss, cached], dtype=np.float64)


def _unpack_rs(prefix, z, rs):
    if prefix + "_keys" in z.files:
        pos, has_gauss, cached = z[prefix + "_meta"]
        rs.set_state(("MT19937", z[prefix + "_keys"], int(pos), int(has_gauss), float(cached)))


def _bn_modules(net):
    return [m for m in net.listModules() if isinstance(m, nn.SpatialBatchNormalization)]


def save(path, S):
    """saveAs (train.lua:252-261) + the optimiser state."""
    out = {"PARAMETERS_G": S.PARAMETERS_G.numpy(), "PARAMETERS_D": S.PARAMETERS_D.numpy(),
           "EPOCH": np.int64(S.EPOCH), "OPT": np.array(json.dumps(S.OPT)),
           "rng": np.array([rng().seed, rng().offset], dtype=np.int64)}
    if rng().dev_base is not None:   # hipGraph replay mode: the stream position is offset + *dev_base
        out["rng_dev_base"] = rng().dev_base.cpu().numpy().astype(np.int64)
    _pack_rs("host_random", S.random, out)
    _pack_rs("dataset_random", dataset._rs, out)
    for i, m in enumerate(_bn_modules(S.MODEL_G)):
        out[f"bnG{i}_mean"], out[f"bnG{i}_var"] = m.running_mean.numpy(), m.running_var.numpy()
    for method, per_net in S.OPTSTATE.items():
        for net, st in per_net.items():
            for k, v in st.items():
                key = f"opt/{method}/{net}/{k}"
                if isinstance(v, Tensor):
                    out[key] = v.numpy()
                elif isinstance(v, torch.Tensor):
                    out[key] = v.cpu().numpy()
                elif isinstance(v, (int, float, bool)):
                    if k == "t" and "t_dev" in st:   # replay mode: the device counter is the step count that was applied
                        v = int(st["t_dev"].item())
                    out[key] = np.array(v)
    np.savez(path, **out)
    return path


def load(path, S):
    """--network resume (train.lua:127-142): parameters, optimiser state, BN statistics, epoch."""
    z = np.load(path, allow_pickle=False)
    S.PARAMETERS_G.copy(z["PARAMETERS_G"])
    S.PARAMETERS_D.copy(z["PARAMETERS_D"])
    S.EPOCH = int(z["EPOCH"])
    r = rng()
    r.seed, r.offset = int(z["rng"][0]), int(z["rng"][1])
    if "rng_dev_base" in z.files:
        r.enable_device_base().copy_(torch.from_numpy(z["rng_dev_base"]))
    _unpack_rs("host_random", z, S.random)
    _unpack_rs("dataset_random", z, dataset._rs)
    for i, m in enumerate(_bn_modules(S.MODEL_G)):
        m.running_mean.copy(z[f"bnG{i}_mean"])
        m.running_var.copy(z[f"bnG{i}_var"])
    for key in z.files:
        if not key.startswith("opt/"):
            continue
        _, method, net, k = key.split("/")
        v = z[key]
        st = S.OPTSTATE.setdefault(method, {}).setdefault(net, {})
        if k == "t_dev":
            st[k] = torch.from_numpy(v.astype(np.int64)).to(S.PARAMETERS_G.t.device)
        elif v.ndim == 0:
            st[k] = v.item()
        else:
            st[k] = Tensor.from_numpy(v, fmt="plain")
    return S
