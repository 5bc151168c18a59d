"""Builds the gfx950 HIP extension (libcatgan_hip.so) in-tree with hipcc.

No torch / pybind in the link: the library is a plain C-ABI shared object
(include/catgan.h) so that LuaJIT's FFI, ctypes or a C++ driver can bind it.
Each source is compiled to its own object (in parallel, only when stale) and the
objects are linked together.  comm.hip (RCCL collectives) needs rccl.h at build time only;
librccl.so.1 is bound with dlopen when the first cg_comm_* call is made.
"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libcatgan_hip.so")
SOURCES = ["gemm.hip", "winograd.hip", "skinny.hip", "wino3.hip", "headwg.hip", "ops.hip", "fused.hip", "comm.hip", "locnet.hip", "net.hip"]
ARCH = "gfx950"
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"]   # no -munsafe-fp-atomics: nothing on the path adds floats atomically
# extra -D switches of the instrumented builds: CG_BUILD_DEFINES="-DCG_TRACE" (per-workgroup timestamps, scripts/wg_trace.py) or
# "-DCG_TIMING_PROBE=<mask>" (timing-only: halved K loops per GEMM family, csrc/common.h; such a library reports ABI version -1 and the
# hosts refuse it unless CG_ALLOW_TIMING_PROBE=1 - bench.py then labels its line).  A change of the value forces a full rebuild.
FLAGS += os.environ.get("CG_BUILD_DEFINES", "").split()


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "catgan.h")]   # net_ktable.inc is regenerated from catgan.h


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _defines_changed():
    """the -D switches the objects were built with (lib/obj/defines.txt) differ from this build's"""
    tag = os.path.join(OBJ_DIR, "defines.txt")
    cur = os.environ.get("CG_BUILD_DEFINES", "")
    return (open(tag).read() if os.path.exists(tag) else "") != cur


def _stale():
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    return _newer(LIB_PATH, srcs + _headers()) or _defines_changed()


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or os.path.join(ROCM, "bin", "hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    if _defines_changed():
        force = True
        open(os.path.join(OBJ_DIR, "defines.txt"), "w").write(os.environ.get("CG_BUILD_DEFINES", ""))
    _generate("gen_net_ktable.py", os.path.join(CSRC, "net_ktable.inc"))   # the launch table net.hip dispatches through
    hdrs = _headers()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(OBJ_DIR, s.replace(".hip", ".o"))
        jobs.append((src, obj, force or _newer(obj, [src] + hdrs)))

    def compile_one(job):
        src, obj, need = job
        if need:
            cmd = [hipcc, *FLAGS, f"-I{os.path.join(ROCM, 'include')}", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        objs = list(ex.map(compile_one, jobs))
    link = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB_PATH]
    link += ["-ldl"]   # comm.hip binds RCCL with dlopen at run time: no link-time dependency on librccl
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    build_tools(hipcc, verbose)
    return LIB_PATH


def _generate(script, out_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location(script[:-3], os.path.join(HERE, "..", "scripts", script))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    return gen.main(out_path)


def build_tools(hipcc, verbose=False):
    """tools/abi_replay: the interpreter-free host that replays a recorded call sequence (tests/test_abi_step.py)."""
    import importlib.util
    root = os.path.join(HERE, "..")
    spec = importlib.util.spec_from_file_location("gen_abi_dispatch", os.path.join(root, "scripts", "gen_abi_dispatch.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gen.main(os.path.join(root, "tools", "abi_dispatch.inc"))
    cmd = [hipcc, "-O2", "-std=c++17", "-x", "c++", os.path.join(root, "tools", "abi_replay.cpp"), "-o", os.path.join(root, "tools", "abi_replay"),
           f"-L{LIB_DIR}", "-lcatgan_hip", f"-Wl,-rpath,{LIB_DIR}"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    # stand-alone micro-benchmarks (tools/*.hip): MFMA ceiling, LDS -> MFMA loop, LDS-direct-load semantics
    tools = os.path.join(root, "tools")
    jobs = [("mfma_peak.hip", "mfma_peak", []), ("lds_mfma.hip", "lds_mfma", []), ("lds_mfma.hip", "lds_mfma_agpr", ["-DACC_AGPR"]),
            ("glds_test.hip", "glds_test", [])]

    def one(job):
        src, out, extra = (os.path.join(tools, job[0]), os.path.join(tools, job[1]), job[2])
        if os.path.exists(src) and _newer(out, [src]):
            c = [hipcc, f"--offload-arch={ARCH}", "-O3", "-Wno-unused-result", "-Wno-unused-value", *extra, src, "-o", out]
            if verbose:
                print(" ".join(c))
            subprocess.check_call(c, stderr=subprocess.DEVNULL)

    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        list(ex.map(one, jobs))


if __name__ == "__main__":
    print(build(force=True, verbose=True))
