"""Torch7 serialisation: cat-generator_amd/t7.py against a byte string assembled by hand from the format
description (torch7 File.lua, binary mode), back-references, and the export / import of the reference's checkpoint table
{D, G, opt, plot_data, epoch} with the nets as nn / cudnn / stn objects (train.lua:252-261, 127-142)."""
import importlib
import io
import os
import struct

import numpy as np

cg = importlib.import_module("cat-generator_amd")
t7 = importlib.import_module("cat-generator_amd.t7")


def _i(v):
    return struct.pack("<i", v)


def _q(v):
    return struct.pack("<q", v)


def _s(s):
    return _i(len(s)) + s.encode()


def test_byte_layout_of_a_table_with_a_tensor():
    """{ [1] = 2.5, name = "ab", t = FloatTensor{{1,2,3},{4,5,6}}, ok = true }, spelled out field by field."""
    tensor = (_i(4) + _i(2) + _s("V 1") + _s("torch.FloatTensor") + _i(2) + _q(2) + _q(3) + _q(3) + _q(1) + _q(1)
              + _i(4) + _i(3) + _s("V 1") + _s("torch.FloatStorage") + _q(6) + np.arange(1, 7, dtype="<f4").tobytes())
    blob = (_i(3) + _i(1) + _i(4)                                   # table, index 1, four entries
            + _i(1) + struct.pack("<d", 1.0) + _i(1) + struct.pack("<d", 2.5)
            + _i(2) + _s("name") + _i(2) + _s("ab")
            + _i(2) + _s("t") + tensor
            + _i(2) + _s("ok") + _i(5) + _i(1))
    got = t7.Reader(io.BytesIO(blob)).read()
    assert got[1] == 2.5 and got["name"] == "ab" and got["ok"] is True
    np.testing.assert_array_equal(got["t"], np.arange(1, 7, dtype=np.float32).reshape(2, 3))
    out = io.BytesIO()
    t7.Writer(out).write({1: 2.5, "name": "ab", "t": np.arange(1, 7, dtype=np.float32).reshape(2, 3), "ok": True})
    assert out.getvalue() == blob


def test_strided_tensor_views_and_legacy_class_header():
    """A transposed view (strides 1, 3 over a 6-element storage, offset 1) and the pre-versioning header (class name
    without the "V 1" string) as older torch.save files carry."""
    blob = (_i(4) + _i(1) + _s("torch.DoubleTensor") + _i(2) + _q(3) + _q(2) + _q(1) + _q(3) + _q(1)
            + _i(4) + _i(2) + _s("torch.DoubleStorage") + _q(6) + np.arange(6, dtype="<f8").tobytes())
    got = t7.Reader(io.BytesIO(blob)).read()
    np.testing.assert_array_equal(got, np.arange(6, dtype=np.float64).reshape(2, 3).T)


def test_back_references_objects_storages_and_empty_tensors(tmp_path):
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    mod = t7.TorchObject("nn.Linear", {"weight": a, "bias": np.zeros(3, np.float32), "train": False,
                                      "size": t7.Storage(np.array([3, 4], np.int64)), "output": np.zeros((0,), np.float32)})
    obj = {"first": mod, "again": mod, "cuda": t7.CudaTensor(a), "list": [1, "two", None, 4.5], "n": -7}
    z = t7.load(t7.save(str(tmp_path / "x.t7"), obj))
    assert z["first"] is z["again"] and z["first"].typename == "nn.Linear"
    np.testing.assert_array_equal(z["first"]["weight"], a)
    np.testing.assert_array_equal(z["cuda"], a)
    assert z["first"]["train"] is False and z["first"]["output"].size == 0 and z["n"] == -7
    np.testing.assert_array_equal(z["first"]["size"], [3, 4])
    assert t7.table_list(z["list"]) == [1, "two"] and z["list"][4] == 4.5   # a nil ends the array part, as in Lua
    raw = open(str(tmp_path / "x.t7"), "rb").read()
    assert raw.count(b"nn.Linear") == 1 and raw.count(b"torch.CudaStorage") == 1


def test_checkpoint_table_export_and_import(tmp_path):
    """export_t7 -> import_t7: same module trees (class names in listModules order), identical flat parameter vectors,
    batch-norm running statistics, evaluate/training flags, OPT / epoch / plot_data / optstate."""
    cg.manual_seed(3)
    G, D = cg.models.create_G((3, 32, 32), 100), cg.models.create_D((3, 32, 32))
    S = cg.adversarial.State(dict(batchSize=8, D_clamp=0.5), G, D)
    S.EPOCH = 5
    bn = [m for m in G.listModules() if isinstance(m, cg.nn.SpatialBatchNormalization)]
    bn[0].running_mean.copy(np.linspace(-1, 1, bn[0].running_mean.nElement()).astype(np.float32))
    bn[0].running_var.copy(np.linspace(0.5, 2, bn[0].running_var.nElement()).astype(np.float32))
    n = S.PARAMETERS_D.nElement()
    S.OPTSTATE["adam"]["D"].update(t=3, m=cg.Tensor.from_numpy(np.full(n, 0.25, np.float32)), v=cg.Tensor.from_numpy(np.full(n, 0.5, np.float32)))
    D.evaluate()
    path = cg.checkpoint.export_t7(str(tmp_path / "adversarial.net"), S, plot_data=[[1, 0.5, 0.6, 0.4]])
    z = cg.checkpoint.import_t7(path)
    assert z["epoch"] == 5 and z["plot_data"] == [[1, 0.5, 0.6, 0.4]] and z["opt"]["D_clamp"] == 0.5 and z["opt"]["batchSize"] == 8
    for a, b in ((G, z["G"]), (D, z["D"])):
        assert [m.typename for m in a.listModules()] == [m.typename for m in b.listModules()]
    pg, _ = z["G"].getParameters()
    pd, _ = z["D"].getParameters()
    np.testing.assert_array_equal(pg.numpy(), S.PARAMETERS_G.numpy())
    np.testing.assert_array_equal(pd.numpy(), S.PARAMETERS_D.numpy())
    bn2 = [m for m in z["G"].listModules() if isinstance(m, cg.nn.SpatialBatchNormalization)]
    np.testing.assert_array_equal(bn2[0].running_var.numpy(), bn[0].running_var.numpy())
    np.testing.assert_array_equal(bn2[0].running_mean.numpy(), bn[0].running_mean.numpy())
    assert not any(m.train for m in z["D"].listModules()) and all(m.train for m in z["G"].listModules())
    st = z["optstate"]["adam"]["D"]
    assert st["t"] == 3 and st["m"].shape == (n,) and float(st["v"][0]) == 0.5
    # the file is a plain torch.save table: the nets are torch objects with the reference's class names
    raw = t7.load(path)
    assert raw["G"].typename == "nn.Sequential" and raw["D"].typename == "nn.Sequential"
    names = {m.typename for m in G.listModules()} | {m.typename for m in D.listModules()}
    assert {"cudnn.SpatialConvolution", "nn.SpatialBatchNormalization", "nn.BilinearSamplerBHWD", "nn.AffineGridGeneratorBHWD"} <= names
    conv = t7.table_list(raw["G"]["modules"])[4]
    assert conv.typename == "cudnn.SpatialConvolution" and conv["weight"].shape == (512, 512, 3, 3) and conv["output"].size == 0
    assert os.path.getsize(path) > 4 * 2 * (S.PARAMETERS_G.nElement() + n)   # parameters and (zero) gradients, as torch.save writes


def test_byte_fixture_of_a_flattened_nested_sequential():
    """What torch.save writes for a small net AFTER getParameters() (train.lua:184-185 flattens before anything is saved): every
    weight / bias is a view of ONE torch.FloatStorage (and every gradWeight / gradBias of one other), so the file holds each
    storage once - at its first use - and back-references (the bare object index) afterwards.  The bytes are assembled here from
    the documented layout (File.lua writeObject, binary mode; nn classes have no write() of their own, so an instance is its class
    header `V 1`, its name, and ONE table with its fields):

        nn.Sequential { train = true, modules = { nn.Linear(3 -> 2){ weight, bias, gradWeight, gradBias },
                                                  nn.Sequential { modules = { nn.PReLU{ weight, gradWeight }, nn.Sigmoid{} } } } }
        parameters  : FloatStorage P of 9 floats  = [ W(2x3) | b(2) | alpha(1) ]      -> views at offsets 1, 7, 9
        gradients   : FloatStorage Gs of 9 floats (zeros)                              -> the same offsets

    Reading it gives the module tree with the right values (views resolved through strides / offsets / the shared storage), and
    the engine imports it (t7_nn.from_t7) as nn modules whose flat parameter vector IS storage P."""
    P = np.array([0.5, -1.0, 2.0, 1.5, 0.25, -0.75, 0.1, -0.2, 0.25], "<f4")
    Gs = np.zeros(9, "<f4")
    idx = iter(range(1, 100))

    def torch_obj(name, payload):
        return _i(4) + _i(next(idx)) + _s("V 1") + _s(name) + payload

    def table(entries):                       # entries: list of (key bytes, value bytes)
        return _i(3) + _i(next(idx)) + _i(len(entries)) + b"".join(k + v for k, v in entries)

    key = lambda s: _i(2) + _s(s)
    num = lambda v: _i(1) + struct.pack("<d", v)
    storage_ids = {}

    def tensor(sizes, offset, which, data):
        body = _i(len(sizes)) + b"".join(_q(s) for s in sizes)
        strides, st = [], 1
        for s in reversed(sizes):
            strides.append(st); st *= s
        body += b"".join(_q(s) for s in reversed(strides)) + _q(offset)
        head = _i(4) + _i(next(idx)) + _s("V 1") + _s("torch.FloatTensor")
        if which in storage_ids:                                   # a back-reference: type tag + index, nothing else
            return head + body + _i(4) + _i(storage_ids[which])
        sid = next(idx)
        storage_ids[which] = sid
        return head + body + _i(4) + _i(sid) + _s("V 1") + _s("torch.FloatStorage") + _q(data.size) + data.tobytes()

    # objects are numbered in order of first appearance: Sequential(1) fields(2) modules(3) Linear(4) fields(5) weight(6) P(7) ...
    seq_i = next(idx); seq_f = next(idx)
    def linear():
        li, lf = next(idx), next(idx)
        ent = [(key("weight"), tensor([2, 3], 1, "P", P)), (key("bias"), tensor([2], 7, "P", P)),
               (key("gradWeight"), tensor([2, 3], 1, "G", Gs)), (key("gradBias"), tensor([2], 7, "G", Gs))]
        return _i(4) + _i(li) + _s("V 1") + _s("nn.Linear") + _i(3) + _i(lf) + _i(len(ent)) + b"".join(k + v for k, v in ent)
    def inner():
        si, sf = next(idx), next(idx)
        mi = next(idx)
        pi, pf = next(idx), next(idx)
        pre = [(key("weight"), tensor([1], 9, "P", P)), (key("gradWeight"), tensor([1], 9, "G", Gs)), (key("nOutputPlane"), num(0))]
        prelu = _i(4) + _i(pi) + _s("V 1") + _s("nn.PReLU") + _i(3) + _i(pf) + _i(len(pre)) + b"".join(k + v for k, v in pre)
        gi, gf = next(idx), next(idx)
        sig = _i(4) + _i(gi) + _s("V 1") + _s("nn.Sigmoid") + _i(3) + _i(gf) + _i(0)
        mods = _i(3) + _i(mi) + _i(2) + num(1) + prelu + num(2) + sig
        return _i(4) + _i(si) + _s("V 1") + _s("nn.Sequential") + _i(3) + _i(sf) + _i(1) + key("modules") + mods
    mods_i = next(idx)
    modules = _i(3) + _i(mods_i) + _i(2) + num(1) + linear() + num(2) + inner()
    blob = (_i(4) + _i(seq_i) + _s("V 1") + _s("nn.Sequential") + _i(3) + _i(seq_f) + _i(2)
            + key("train") + _i(5) + _i(1) + key("modules") + modules)
    assert blob.count(b"torch.FloatStorage") == 2 and blob.count(b"torch.FloatTensor") == 6      # two storages, six views

    net = t7.Reader(io.BytesIO(blob)).read()
    assert net.typename == "nn.Sequential" and net["train"] is True
    lin, sub = t7.table_list(net["modules"])
    assert lin.typename == "nn.Linear" and sub.typename == "nn.Sequential"
    np.testing.assert_array_equal(lin["weight"], P[0:6].reshape(2, 3))
    np.testing.assert_array_equal(lin["bias"], P[6:8])
    prelu, sig = t7.table_list(sub["modules"])
    assert prelu.typename == "nn.PReLU" and sig.typename == "nn.Sigmoid" and sig.fields == {}
    np.testing.assert_array_equal(prelu["weight"], P[8:9])
    np.testing.assert_array_equal(lin["gradWeight"], np.zeros((2, 3), np.float32))
    # into the engine: the same tree, and its flat parameter vector is storage P (Torch7's depth-first order, weight then bias)
    t7_nn = importlib.import_module("cat-generator_amd.t7_nn")
    m = t7_nn.from_t7(net)
    assert [x.typename for x in m.listModules()] == ["nn.Sequential", "nn.Linear", "nn.Sequential", "nn.PReLU", "nn.Sigmoid"]
    flat, _ = m.getParameters()
    np.testing.assert_array_equal(flat.numpy(), P)
    # and back out: our writer emits the same classes with the same fields (storages per tensor: Torch7 accepts either)
    out = io.BytesIO()
    t7.Writer(out).write(t7_nn.to_t7(m))
    again = t7.Reader(io.BytesIO(out.getvalue())).read()
    np.testing.assert_array_equal(t7.table_list(again["modules"])[0]["weight"], P[0:6].reshape(2, 3))
    np.testing.assert_array_equal(t7.table_list(t7.table_list(again["modules"])[1]["modules"])[0]["weight"].reshape(-1), P[8:9])
