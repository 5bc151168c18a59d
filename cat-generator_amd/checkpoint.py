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
    kind, keys, pos, has_gauss, cached = rs if isinstance(rs, tuple) else rs.get_state()
    out[prefix + "_keys"] = np.asarray(keys, dtype=np.uint32)
    out[prefix + "_meta"] = np.array([pos, has_gauss, cached], dtype=np.float64)


def _unpack_rs(prefix, z, rs):
    if prefix + "_keys" in z.files:
        pos, has_gauss, cached = z[prefix + "_meta"]
        state = ("MT19937", z[prefix + "_keys"], int(pos), int(has_gauss), float(cached))
        rs(state) if callable(rs) else rs.set_state(state)


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
    _pack_rs("dataset_random", dataset.checkpoint_state(), out)   # before a pending AsyncLoader prefetch
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
    _unpack_rs("dataset_random", z, dataset.restore_state)
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


# ------------------------------------------------------------------ Torch7-format export / import
def export_t7(path, S, plot_data=None, cuda=False, normalize_mean=None, normalize_std=None):
    """saveAs (train.lua:252-261) in torch.save's own format: {D, G, opt, plot_data, epoch, normalize_mean, normalize_std},
    the nets as nn / cudnn / stn objects with canonical parameter layouts.  Also writes `optstate` - the field the reference's
    resume reads (train.lua:132) but its saveAs never wrote - with optim.adam's t / m / v over the flat vectors."""
    from . import t7, t7_nn
    opt = {k: v for k, v in S.OPT.items() if isinstance(v, (int, float, str, bool))}
    optstate = {}
    for method, per_net in S.OPTSTATE.items():
        optstate[method] = {}
        for net, st in per_net.items():
            d = {}
            for k, v in st.items():
                if isinstance(v, Tensor):
                    d[k] = np.ascontiguousarray(v.numpy(), dtype=np.float32)
                elif isinstance(v, (int, float, bool)) and k != "device_step":
                    d[k] = int(st["t_dev"].item()) if k == "t" and "t_dev" in st else v
            optstate[method][net] = d
    obj = {"D": t7_nn.to_t7(S.MODEL_D, cuda), "G": t7_nn.to_t7(S.MODEL_G, cuda), "opt": opt,
           "plot_data": [list(r) for r in (plot_data or [])], "epoch": int(S.EPOCH), "optstate": optstate}
    if normalize_mean is not None:
        obj["normalize_mean"], obj["normalize_std"] = normalize_mean, normalize_std
    return t7.save(path, obj)


def import_t7(path):
    """torch.load of a {D, G, opt, epoch, ...} file (train.lua:129-135, sample.lua:69-75): returns a dict with the nets
    rebuilt as engine modules (G, D), and opt / epoch / plot_data / optstate as plain Python values."""
    from . import t7, t7_nn
    z = t7.load(path)
    out = {k: v for k, v in z.items() if k not in ("D", "G")}
    for k in ("D", "G"):
        if z.get(k) is not None:
            out[k] = t7_nn.from_t7(z[k])
    if isinstance(out.get("plot_data"), dict):
        out["plot_data"] = [t7.table_list(r) if isinstance(r, dict) else r for r in t7.table_list(out["plot_data"])]
    return out


def load_t7(path, S):
    """--network with a torch.save file (train.lua:127-142): parameters and batch-norm statistics of D and G, epoch, and the
    optimiser state when the file has one (the reference's own files do not: Adam then restarts, as it does upstream)."""
    z = import_t7(path)
    for net, flat in (("G", S.PARAMETERS_G), ("D", S.PARAMETERS_D)):
        src, _ = z[net].getParameters()
        if src.nElement() != flat.nElement():
            raise ValueError(f"{path}: net {net} has {src.nElement()} parameters, the configured model {flat.nElement()}")
        flat.copy(src.numpy())
    for a, b in zip(_bn_modules(S.MODEL_G), _bn_modules(z["G"])):
        a.running_mean.copy(b.running_mean.numpy()); a.running_var.copy(b.running_var.numpy())
    S.EPOCH = int(z.get("epoch", 0)) + 1                    # train.lua:133
    for method, per_net in (z.get("optstate") or {}).items():
        for net, st in per_net.items():
            dst = S.OPTSTATE.setdefault(method, {}).setdefault(net, {})
            for k, v in st.items():
                dst[k] = Tensor.from_numpy(v, fmt="plain") if isinstance(v, np.ndarray) else v
    return S
