"""Records the C-ABI calls of a piece of host code for tools/abi_replay (see its header for the trace format).

    rec = Recorder(cg)                 # cg = the cat-generator_amd package
    rec.start()                        # device memory image (every allocator segment) -> blob, tracing on
    ... host code: only ABI calls may touch device memory from here on ...
    rec.stop(dumps={"pD.bin": (ptr, nbytes), ...}, trace_path, blob_path)

Device pointers are translated into (allocator segment, byte offset), so the replayer can rebuild the same memory image
with cg_malloc and needs neither PyTorch nor the original addresses.  Test tooling: nothing in the product imports it."""
import ctypes
import importlib

import torch

PTR_TYPES = {"void*", "const void*", "float*", "const float*", "double*", "const double*", "int32_t*", "const int32_t*",
             "uint64_t*", "const uint64_t*"}
PARR_TYPES = {"const float* const*", "float* const*"}
HOST_ONLY = {"cg_stream_on_queue"}   # a side stream on a chosen hardware queue (adversarial._side_stream): scheduling, not data
NET_OUT = {"id", "draws", "y", "ynd", "ydims", "yfmt", "gx", "gnd", "gdims", "gfmt", "nbuckets"}   # scripts/gen_abi_dispatch.py


class Recorder:
    def __init__(self, cg):
        self.cg, self.lib = cg, cg.lib()
        self.segs = []        # (address, size, has_initial_contents)
        self.blob = bytearray()
        self.blob_off = []
        self.vals = []        # (call index, address, bytes): tensors returned by cg_net_forward / cg_net_backward
        self.handles = {}     # cg_net handle value -> index

    def _segments(self):
        return sorted((s["address"], s["total_size"]) for s in torch.cuda.memory_snapshot())

    def start(self):
        torch.cuda.synchronize()
        st = self.cg.tensor.stream()
        for addr, size in self._segments():
            buf = (ctypes.c_char * size)()
            self.lib.memcpy_d2h(st, buf, addr, size)
            torch.cuda.synchronize()
            self.segs.append((addr, size, True))
            self.blob_off.append(len(self.blob))
            self.blob += bytes(buf)
        self.lib.trace = []

    def _seg_of(self, p):
        for k, addr, size in self.vals:      # a tensor a planned pass returned: reached through that call in the replay
            if addr <= p < addr + size:
                return f"v:{k}:{p - addr}"
        for i, (addr, size, _) in enumerate(self.segs):
            if addr <= p < addr + size:
                return f"p:{i}:{p - addr}"
        raise ValueError(f"device pointer {p:#x} is in no allocator segment")

    def _ptr(self, v):
        if v is None:
            return "n"
        if hasattr(v, "value"):      # ctypes.c_void_p
            v = v.value
            if v is None:
                return "n"
        return self._seg_of(int(v))

    def _handle(self, v):
        v = v.value if hasattr(v, "value") else int(v)
        if v not in self.handles:
            self.handles[v] = len(self.handles)
        return self.handles[v]

    def _net_args(self, name, args, index):
        """Argument tokens of a cg_net_* / cg_graph_* call (handles, host arrays, out-parameters by name)."""
        out = []
        protos = self.lib.protos[name][1]
        for (typ, an), v in zip(protos, args):
            if an in ("net", "graph_exec"):
                out.append(f"h:{self._handle(v)}")
            elif an == "stream":
                out.append("s")
            elif an in NET_OUT and typ.endswith("*"):
                out.append("o")
            elif typ == "const long*":
                out.append("L:" + ",".join(str(int(e)) for e in v))
            elif typ == "const char*":
                out.append("c:" + (v.decode() if isinstance(v, bytes) else str(v)))
            elif typ in PTR_TYPES:
                out.append(self._ptr(v))
            elif typ in ("float", "double"):
                out.append("f:" + float(v).hex())
            elif typ in ("size_t", "uint64_t"):
                out.append(f"u:{int(v)}")
            else:
                out.append(f"i:{int(v)}")
        RET = {"cg_net_forward": (10, 11, 12), "cg_net_backward": (7, 8, 9), "cg_net_forward_pair": (14, 15, 16), "cg_net_pair_join": (2, 3, 4)}
        if name in RET:   # the tensor this call returned
            yi, ni, di = RET[name]
            ptr, nd, dims = args[yi]._obj.value, args[ni]._obj.value, args[di]
            n = 1
            for k in range(nd):
                n *= int(dims[k])
            self.vals.append((index, ptr, n * 4))
        return out

    def stop(self, dumps, trace_path, blob_path):
        torch.cuda.synchronize()
        calls, self.lib.trace = self.lib.trace, None
        known = {a for a, _, _ in self.segs}
        for addr, size in self._segments():      # segments the recorded code made the allocator create: scratch, start zeroed
            if addr not in known:
                self.segs.append((addr, size, False))
                self.blob_off.append(-1)
        lines = [f"seg|{i}|{size}|{self.blob_off[i]}" for i, (_, size, _) in enumerate(self.segs)]
        protos = self.lib.protos
        # nets built before the recording started: their description first (planned.PlannedNet.prologue)
        P = importlib.import_module(self.cg.__name__ + ".planned")
        used = {(a[0].value if hasattr(a[0], "value") else int(a[0])) for n_, a in calls if n_.startswith("cg_net_")}
        ncall = 0
        for net in sorted(P.LIVE, key=lambda n_: n_.h.value):
            if net.h.value not in used:
                continue
            hk = self._handle(net.h)
            for name, toks in net.prologue():
                toks = [f"H:{hk}" if t == "H" else f"h:{hk}" if t == "h" else self._ptr(t[1]) if isinstance(t, tuple) else t for t in toks]
                lines.append("|".join(["call", name] + toks))
                ncall += 1
        self.prologue_calls = ncall
        for name, args in calls:
            if name in HOST_ONLY:        # services with host out-parameters that move no data: the replayer has its one stream
                continue
            if name.startswith(("cg_net_", "cg_graph_")):
                lines.append("|".join(["call", name] + self._net_args(name, args, ncall)))
                ncall += 1
                continue
            ncall += 1
            out = ["call", name]
            for (typ, argname), v in zip(protos[name][1], args):
                if typ == "void*" and argname == "stream":
                    out.append("s")
                elif typ in PTR_TYPES:
                    out.append(self._ptr(v))
                elif typ in PARR_TYPES:
                    out.append("n" if v is None else f"a:{len(v)}:" + ",".join(self._ptr(e) for e in v))
                elif typ in ("float", "double"):
                    out.append("f:" + float(v).hex())
                elif typ in ("size_t", "uint64_t"):
                    out.append(f"u:{int(v)}")
                elif typ in ("int", "long", "int32_t"):
                    out.append(f"i:{int(v)}")
                else:
                    raise ValueError(f"{name}: argument type {typ} cannot be replayed")
            lines.append("|".join(out))
        for fname, (ptr, nbytes) in dumps.items():
            lines.append(f"dump|{self._seg_of(int(ptr))}|{int(nbytes)}|{fname}")
        open(trace_path, "w").write("\n".join(lines) + "\n")
        open(blob_path, "wb").write(self.blob)
        return ncall
