"""Records the C-ABI calls of a piece of host code for tools/abi_replay (see its header for the trace format).

    rec = Recorder(cg)                 # cg = the cat-generator_amd package
    rec.start()                        # device memory image (every allocator segment) -> blob, tracing on
    ... host code: only ABI calls may touch device memory from here on ...
    rec.stop(dumps={"pD.bin": (ptr, nbytes), ...}, trace_path, blob_path)

Device pointers are translated into (allocator segment, byte offset), so the replayer can rebuild the same memory image
with cg_malloc and needs neither PyTorch nor the original addresses.  Test tooling: nothing in the product imports it."""
import ctypes

import torch

PTR_TYPES = {"void*", "const void*", "float*", "const float*", "double*", "const double*", "int32_t*", "const int32_t*",
             "uint64_t*", "const uint64_t*"}
PARR_TYPES = {"const float* const*", "float* const*"}


class Recorder:
    def __init__(self, cg):
        self.cg, self.lib = cg, cg.lib()
        self.segs = []        # (address, size, has_initial_contents)
        self.blob = bytearray()
        self.blob_off = []

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
        for name, args in calls:
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
        return len(calls)
