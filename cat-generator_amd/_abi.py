"""ctypes binding of include/catgan.h.

Prototypes are parsed from the header itself, so the Python host, the LuaJIT
`ffi.cdef` and the library cannot drift apart.  There is no
fallback: if libcatgan_hip.so is missing or a call fails, this raises.
"""
import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "catgan.h")
LIB_PATH = os.environ.get("CATGAN_LIB") or os.path.join(HERE, "lib", "libcatgan_hip.so")   # CATGAN_LIB: A/B another build

_CTYPES = {
    "void*": C.c_void_p, "const void*": C.c_void_p, "void**": C.POINTER(C.c_void_p),
    "float*": C.c_void_p, "const float*": C.c_void_p,
    "const float* const*": C.c_void_p, "float* const*": C.c_void_p,
    "double*": C.c_void_p, "const double*": C.c_void_p, "const unsigned char*": C.c_void_p,
    "int32_t*": C.c_void_p, "const int32_t*": C.c_void_p, "uint64_t*": C.c_void_p, "const uint64_t*": C.c_void_p,
    "int32_t": C.c_int32, "int*": C.POINTER(C.c_int), "long*": C.POINTER(C.c_long), "const int*": C.POINTER(C.c_int),
    "int": C.c_int, "long": C.c_long, "float": C.c_float, "double": C.c_double,
    "size_t": C.c_size_t, "uint64_t": C.c_uint64, "const char*": C.c_char_p, "void": None,
    "const long*": C.POINTER(C.c_long), "float**": C.POINTER(C.c_void_p), "char*": C.c_void_p, "size_t*": C.POINTER(C.c_size_t),
    "cg_alloc_fn": C.c_void_p, "cg_hook_fn": C.c_void_p,
}
ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)                                          # cg_alloc_fn
HOOK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)    # cg_hook_fn


class CatganError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype_str, [(argtype_str, argname), ...])} for every function the header declares."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(cg_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        ret = " ".join(ret.replace("*", "* ").split()).replace(" *", "*").strip()
        ret = ret.replace("* ", "*")
        arglist = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a)
                typ = mm.group(1).strip().replace(" *", "*")
                arglist.append((typ, mm.group(2)))
        protos[name] = (ret, arglist)
    return protos


class Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise CatganError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        self._dll = C.CDLL(LIB_PATH)
        self.trace = None
        self.protos = parse_header()
        for name, (ret, args) in self.protos.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError as e:
                raise CatganError(f"libcatgan_hip.so does not export {name} declared in catgan.h") from e
            fn.restype = _CTYPES[ret]
            fn.argtypes = [_CTYPES[t] for t, _ in args]
            if ret == "int" and name != "cg_abi_version" and not name.endswith("_supported"):   # predicates return 0 / 1, not a status
                setattr(self, name[3:], self._checked(fn, name))
            else:
                setattr(self, name[3:], fn)
        v = self.abi_version()
        self.timing_probe = v == -1 and os.environ.get("CG_ALLOW_TIMING_PROBE") == "1"
        if v != 1 and not self.timing_probe:
            raise CatganError("ABI version mismatch" + (" (a -DCG_TIMING_PROBE build: wrong results by construction; CG_ALLOW_TIMING_PROBE=1 "
                                                        "lets bench.py time it)" if v == -1 else ""))

    def _checked(self, fn, name):
        last_error = self._dll.cg_last_error
        last_error.restype = C.c_char_p

        lib = self

        def call(*a):
            if lib.trace is not None:      # tools/abi_replay: every state-changing entry point with its arguments
                lib.trace.append((name, a))
            rc = fn(*a)
            if rc != 0:
                raise CatganError(f"{name}: {last_error().decode()}")
            return 0

        call.__name__ = name
        return call


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = Lib()
    return _lib
