#!/usr/bin/env python
"""Static checks of the LuaJIT layer (lua/) - there is no Lua interpreter in the build image.

1. every `C.cg_*` reference names a function include/catgan.h declares, and every CALL passes exactly the number of
   arguments the prototype has (LuaJIT's FFI raises "wrong number of arguments" only at run time);
2. block structure: function / if / for / while / repeat / do ... end / until balance per file (strings, long brackets and
   comments skipped), so a missing `end` cannot hide;
3. every class the reference's model / loop files instantiate is defined (nn.*, cudnn.*, optim.*).
Exit code 0 = clean.  Run by tests/test_abi_and_host.py."""
import importlib.util
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def protos():
    spec = importlib.util.spec_from_file_location("_abi", os.path.join(ROOT, "cat-generator_amd", "_abi.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.parse_header(os.path.join(ROOT, "include", "catgan.h"))


def strip_lua(src):
    """Replace comments and string contents by spaces (same length, newlines kept)."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("--", i):
            m = re.match(r"--\[(=*)\[", src[i:])
            if m:
                close = "]" + m.group(1) + "]"
                j = src.find(close, i)
                j = n if j < 0 else j + len(close)
            else:
                j = src.find("\n", i)
                j = n if j < 0 else j
            out.append(re.sub(r"[^\n]", " ", src[i:j])); i = j
        elif c in "\"'":
            j = i + 1
            while j < n and src[j] != c:
                j += 2 if src[j] == "\\" else 1
            out.append(c + " " * (j - i - 1) + c); i = j + 1
        elif c == "[" and re.match(r"\[(=*)\[", src[i:]):
            m = re.match(r"\[(=*)\[", src[i:])
            close = "]" + m.group(1) + "]"
            j = src.find(close, i)
            j = n if j < 0 else j + len(close)
            out.append(re.sub(r"[^\n]", " ", src[i:j])); i = j
        else:
            out.append(c); i += 1
    return "".join(out)


def count_args(src, i):
    """src[i] == '(' -> (number of top-level arguments, index after the matching ')')."""
    depth, args, seen, j = 0, 0, False, i
    while j < len(src):
        c = src[j]
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
            if depth == 0:
                return (args + 1 if seen else 0), j + 1
        elif c == "," and depth == 1:
            args += 1
        elif not c.isspace() and depth >= 1:
            seen = True
        j += 1
    raise ValueError("unbalanced parentheses")


def check_blocks(path, src):
    errs, stack, pending_loop = [], [], False
    for m in re.finditer(r"\b(function|if|for|while|repeat|do|end|until)\b", src):
        w = m.group(1)
        line = src.count("\n", 0, m.start()) + 1
        if w in ("function", "if", "repeat"):
            stack.append((w, line))
        elif w in ("for", "while"):
            stack.append((w, line)); pending_loop = True
        elif w == "do":
            if pending_loop:
                pending_loop = False     # the `do` of a for / while header
            else:
                stack.append((w, line))
        elif w == "end":
            if not stack or stack[-1][0] == "repeat":
                errs.append(f"{path}:{line}: unexpected `end`"); continue
            stack.pop()
        elif w == "until":
            if not stack or stack[-1][0] != "repeat":
                errs.append(f"{path}:{line}: `until` without `repeat`"); continue
            stack.pop()
    errs += [f"{path}:{l}: `{w}` is never closed" for w, l in stack]
    return errs


def check_strings(path, raw):
    """2c. a quoted string may not run over a line end (Lua 5.1), a long bracket / long comment must close."""
    errs, i, n, line = [], 0, len(raw), 1
    while i < n:
        c = raw[i]
        if raw.startswith("--", i):
            m = re.match(r"--\[(=*)\[", raw[i:])
            if m:
                j = raw.find("]" + m.group(1) + "]", i)
                if j < 0:
                    errs.append(f"{path}:{line}: long comment is never closed"); break
                line += raw.count("\n", i, j); i = j + 2 + len(m.group(1))
            else:
                j = raw.find("\n", i); i = n if j < 0 else j
        elif c in "\"'":
            j = i + 1
            while j < n and raw[j] != c:
                if raw[j] == "\n":
                    errs.append(f"{path}:{line}: string is not terminated on its line"); break
                j += 2 if raw[j] == "\\" else 1
            if errs and errs[-1].endswith("on its line"):
                break
            i = j + 1
        elif c == "[" and re.match(r"\[(=*)\[", raw[i:]):
            m = re.match(r"\[(=*)\[", raw[i:])
            j = raw.find("]" + m.group(1) + "]", i)
            if j < 0:
                errs.append(f"{path}:{line}: long bracket is never closed"); break
            line += raw.count("\n", i, j); i = j + 2 + len(m.group(1))
        else:
            if c == "\n":
                line += 1
            i += 1
    return errs


def check_brackets(path, raw, src):
    """2b. ( [ { balance per file on the comment- / string-stripped text, an unterminated string or long bracket (the stripper runs to
    the end of the file then), and characters Lua 5.1 has no token for."""
    errs, stack = [], []
    pairs = {")": "(", "]": "[", "}": "{"}
    for i, c in enumerate(src):
        if c in "([{":
            stack.append((c, src.count("\n", 0, i) + 1))
        elif c in ")]}":
            if not stack or stack[-1][0] != pairs[c]:
                errs.append(f"{path}:{src.count(chr(10), 0, i) + 1}: unmatched `{c}`")
                break
            stack.pop()
        elif c in "`$@!?\\" or (c == "|" ) or (c == "&"):
            errs.append(f"{path}:{src.count(chr(10), 0, i) + 1}: `{c}` is not a Lua 5.1 token")
    errs += [f"{path}:{l}: `{c}` is never closed" for c, l in stack[:3]]
    # a string / long bracket that swallowed the rest of the file: the stripped text then ends in a run of blanks much longer than the raw tail
    tail_raw = raw.rstrip()[-40:]
    if tail_raw and not src.rstrip().endswith(tail_raw.strip()[-1:]) and not raw.rstrip().endswith(("]]", "]=]")) and src.rstrip()[-1:] != tail_raw[-1:]:
        last = src.rstrip()
        if len(raw.rstrip()) - len(last) > 200:
            errs.append(f"{path}: a string or long bracket near line {last.count(chr(10)) + 1} is not terminated")
    # `=` where `==` is needed inside a condition is a syntax error in Lua: if / elseif / while / until ... <single => ... then / do
    for m in re.finditer(r"\b(if|elseif|while|until)\b([^\n]*?)\b(then|do)\b", src):
        cond = m.group(2)
        if re.search(r"(?<![=~<>])=(?!=)", cond) and "function" not in cond:
            errs.append(f"{path}:{src.count(chr(10), 0, m.start()) + 1}: assignment inside a condition")
    return errs


def check_net_builder():
    """4. the planned executor's builder: lua/catgan/net.lua describes a module with the same (kind, iargs, fargs) as the executable
    twin cat-generator_amd/planned.py, and both use the kind numbers of csrc/net.hip's enum (the header documents them)."""
    errs = []
    hip = open(os.path.join(ROOT, "cat-generator_amd", "csrc", "net.hip")).read()
    enum = dict((n, int(v)) for n, v in re.findall(r"K_([A-Z]+) = (\d+)", hip))
    names = {"SEQ": "Sequential", "CONCAT": "Concat", "CONCATTABLE": "ConcatTable", "LINEAR": "Linear", "CONV": "SpatialConvolution", "PRELU": "PReLU",
             "LRELU": "LeakyReLU", "SIGMOID": "Sigmoid", "BN": "SpatialBatchNormalization", "VIEW": "View", "COPY": "Copy", "TRANSPOSE": "Transpose",
             "UPS": "SpatialUpSamplingNearest", "AVGPOOL": "SpatialAveragePooling", "MAXPOOL": "SpatialMaxPooling", "SDROP": "SpatialDropout",
             "DROP": "Dropout", "AFFMAT": "AffineTransformMatrixGenerator", "AFFGRID": "AffineGridGeneratorBHWD", "SAMPLER": "BilinearSamplerBHWD"}
    want = {names[k]: v for k, v in enum.items() if k in names}
    py = open(os.path.join(ROOT, "cat-generator_amd", "planned.py")).read()
    py_kind = dict((n, int(v)) for n, v in re.findall(r"(\w+)=(\d+)", py[py.index("KIND = dict("):py.index(")", py.index("KIND = dict("))]))
    lua = open(os.path.join(ROOT, "lua", "catgan", "net.lua")).read()
    lua_kind = dict((n.split(".")[1], int(v)) for n, v in re.findall(r"\['((?:nn|cudnn)\.\w+)'\] = (\d+)", lua))
    hdr = open(os.path.join(ROOT, "include", "catgan.h")).read()
    for cls, k in want.items():
        if py_kind.get(cls) != k:
            errs.append(f"planned.py: KIND[{cls}] = {py_kind.get(cls)}, csrc/net.hip says {k}")
        if lua_kind.get(cls) != k:
            errs.append(f"lua/catgan/net.lua: KIND[nn.{cls}] = {lua_kind.get(cls)}, csrc/net.hip says {k}")
        if not re.search(r"\b%d (?:nn|nn\|cudnn)\.%s" % (k, cls), hdr):
            errs.append(f"include/catgan.h: cg_net_add's comment does not list kind {k} as {cls}")
    # argument lists: the fields each describe() reads, per kind, must be the same fields in the same order
    def fields(src, lang):
        out = {}
        if lang == "py":
            for m in re.finditer(r'if n == "(\w+)":\n\s+return k, \[([^\]]*)\], \[([^\]]*)\]', src):
                out[KINDNUM(py_kind, m.group(1))] = (re.findall(r"m\.(\w+)", m.group(2)), re.findall(r"m\.(\w+)", m.group(3)))
        else:
            for m in re.finditer(r"if k == (\d+) then return k, \{([^}]*)\}, \{([^}]*)\} end", src):
                out[int(m.group(1))] = (re.findall(r"m\.(\w+)", m.group(2)), re.findall(r"m\.(\w+)", m.group(3)))
        return out
    def KINDNUM(tab, name):
        return tab[name]
    fp, fl = fields(py, "py"), fields(lua, "lua")
    canon = lambda f: [x for x in f if x not in ("shape",)]
    for k in sorted(set(fp) & set(fl)):
        a, b = fp[k], fl[k]
        if k == 3:
            continue   # nn.Linear: weight.shape[1], weight.shape[0] (0-based) vs weight.shape[2], weight.shape[1] (1-based)
        if (canon(a[0]), canon(a[1])) != (canon(b[0]), canon(b[1])):
            errs.append(f"net builders disagree on kind {k}: planned.py reads {a}, net.lua reads {b}")
    if len(set(fp) & set(fl)) < 7:
        errs.append(f"net builders: only {len(set(fp) & set(fl))} kinds could be compared (parser out of date?)")
    return errs


# ---- 5. the reference's host files against the providers in lua/ -------------------------------------------------------------------
REF_FILES = ["adversarial.lua", "models.lua", "train.lua", "weight-init.lua", "utils/nn_utils.lua", "dataset.lua"]
MANIFEST = os.path.join(ROOT, "tests", "golden", "lua_reference_names.json")
LUA_STD = {"math": {"abs", "ceil", "cos", "exp", "floor", "huge", "log", "max", "min", "pi", "pow", "random", "randomseed", "sin", "sqrt"},
           "os": {"clock", "date", "execute", "exit", "getenv", "remove", "rename", "time"},
           "string": {"byte", "char", "find", "format", "gmatch", "gsub", "len", "lower", "match", "rep", "reverse", "sub", "upper"},
           "table": {"concat", "insert", "remove", "sort", "unpack"}, "io": {"close", "flush", "lines", "open", "popen", "read", "write"}}
STRING_METHODS = LUA_STD["string"]
# globals of the reference that hold one of its own modules / objects: name -> the table its functions are defined on
REF_OWN = {"ADVERSARIAL": "adversarial", "DATASET": "dataset", "MODELS": "models", "NN_UTILS": "nn_utils", "TRAIN_DATA": "result",
           "adversarial": "adversarial", "dataset": "dataset", "models": "models", "nn_utils": "nn_utils"}
# globals that are one of OUR providers under another name
ALIAS = {"DISP": "display"}


def reference_names(refdir):
    """What the reference files pull in: {'requires': {module: [file:line]}, 'calls': {'ns.fn': [...]}, 'methods': {name: [...]},
    'defines': [ 'tbl.fn' / 'tbl:fn' defined by the files themselves ]} - identifiers only, no code."""
    req, calls, meth, defines = {}, {}, {}, set()
    for f in REF_FILES:
        raw = open(os.path.join(refdir, f)).read()
        src = strip_lua(raw)
        for m in re.finditer(r"\brequire\b", src):
            q = re.match(r"require\s*\(?\s*['\"]([^'\"]+)['\"]", raw[m.start():])
            if q:
                req.setdefault(q.group(1), []).append(f"{f}:{src.count(chr(10), 0, m.start()) + 1}")
        for m in re.finditer(r"pcall\(\s*require\s*,\s*['\"]", src):
            q = re.match(r"pcall\(\s*require\s*,\s*['\"]([^'\"]+)['\"]", raw[m.start():])
            if q:
                req.setdefault(q.group(1), []).append(f"{f}:{src.count(chr(10), 0, m.start()) + 1}")
        for m in re.finditer(r"(?<![\w.:])([A-Za-z_]\w*)((?:\.\w+)+)\s*[\({]", src):
            calls.setdefault(m.group(1) + m.group(2), []).append(f"{f}:{src.count(chr(10), 0, m.start()) + 1}")
        for m in re.finditer(r":(\w+)\s*[\({]", src):
            meth.setdefault(m.group(1), []).append(f"{f}:{src.count(chr(10), 0, m.start()) + 1}")
        for m in re.finditer(r"\bfunction\s+([A-Za-z_]\w*)([.:])(\w+)", src):
            defines.add(m.group(1) + m.group(2) + m.group(3))
    trim = lambda d: {k: v[:3] for k, v in sorted(d.items())}
    return {"files": REF_FILES, "requires": trim(req), "calls": trim(calls), "methods": trim(meth), "defines": sorted(defines)}


def provider_index():
    """Names the files under lua/ define: functions 'ns.fn' (function ns.fn / ns.fn = / multiple assignment / table-constructor keys),
    methods (function Cls:m / function Cls.m / Cls.m =), classes (class('nn.X')), modules (lua/<path>.lua)."""
    fns, methods, modules = set(), set(), set()
    for d, _, fs in os.walk(os.path.join(ROOT, "lua")):
        for f in fs:
            if not f.endswith(".lua"):
                continue
            path = os.path.join(d, f)
            rel = os.path.relpath(path, os.path.join(ROOT, "lua"))[:-4].replace(os.sep, ".")
            modules.add(rel[:-5] if rel.endswith(".init") else rel)
            raw = open(path).read()
            src = strip_lua(raw)
            for m in re.finditer(r"\bfunction\s+([A-Za-z_]\w*)([.:])(\w+)", src):
                fns.add(m.group(1) + "." + m.group(3)); methods.add(m.group(3))
            for m in re.finditer(r"(?<![\w.])([A-Za-z_]\w*)\.(\w+)\s*(?:=(?!=)|,)", src):      # ns.fn = ... / ns.a, ns.b = ...
                fns.add(m.group(1) + "." + m.group(2)); methods.add(m.group(2))
            for m in re.finditer(r"class\('(nn|cudnn)\.(\w+)'", raw):
                fns.add(m.group(1) + "." + m.group(2))
            # table constructors bound to a name: `local torch = { FloatTensor = ..., }`, `cutorch = { setDevice = ... }`
            for m in re.finditer(r"(?:local\s+)?([A-Za-z_]\w*)\s*=\s*\{", src):
                depth, j = 0, m.end() - 1
                body_start = j + 1
                while j < len(src):
                    if src[j] == "{":
                        depth += 1
                    elif src[j] == "}":
                        depth -= 1
                        if depth == 0:
                            break
                    j += 1
                body = src[body_start:j]
                # top-level keys only
                dd, k0, keys = 0, 0, []
                for k, ch in enumerate(body):
                    if ch in "{(":
                        dd += 1
                    elif ch in "})":
                        dd -= 1
                    elif dd == 0:
                        mm = re.match(r"\s*([A-Za-z_]\w*)\s*=(?!=)", body[k:]) if (k == 0 or body[k - 1] in ",;{\n ") and (k == 0 or not body[k - 1].isalnum()) else None
                        if mm and (k == 0 or body[:k].rstrip().endswith((",", ";")) or body[:k].strip() == ""):
                            keys.append(mm.group(1))
                for key in keys:
                    fns.add(m.group(1) + "." + key)
    return fns, methods, modules


def check_reference(refdir=None):
    import json
    errs = []
    have_ref = refdir and all(os.path.exists(os.path.join(refdir, f)) for f in REF_FILES)
    if have_ref:
        names = reference_names(refdir)
        if os.path.exists(MANIFEST) and json.load(open(MANIFEST)) != names:
            errs.append("tests/golden/lua_reference_names.json is out of date: python scripts/check_lua_binding.py --write-manifest")
    elif os.path.exists(MANIFEST):
        names = json.load(open(MANIFEST))
    else:
        return ["no reference tree and no tests/golden/lua_reference_names.json: cannot check the reference's names"], None
    fns, methods, modules = provider_index()
    own_modules = {f[:-4].replace("/", ".") for f in names["files"]}
    own_defs = set(names["defines"])
    for mod, where in names["requires"].items():
        if mod not in modules and mod not in own_modules:
            errs.append(f"{where[0]}: require '{mod}' has no provider (lua/{mod.replace('.', '/')}.lua)")
    for call, where in names["calls"].items():
        parts = call.split(".")
        root, fn = parts[0], parts[-1]
        if root in LUA_STD:
            if fn not in LUA_STD[root]:
                errs.append(f"{where[0]}: {call} is not standard Lua 5.1")
            continue
        if root in REF_OWN:
            tbl = REF_OWN[root]
            if f"{tbl}.{fn}" not in own_defs and f"{tbl}:{fn}" not in own_defs:
                errs.append(f"{where[0]}: {call}: the reference does not define {tbl}.{fn} itself")
            continue
        if root in ("OPT", "self", "opt", "m", "node", "v", "tmp", "result", "data", "images", "arg", "module", "model", "net", "this"):
            continue        # field reads of values, not namespaces
        ns = ALIAS.get(root, root)
        if f"{ns}.{fn}" not in fns:
            errs.append(f"{where[0]}: {call} has no provider in lua/")
    for m_, where in names["methods"].items():
        if m_ in methods or m_ in STRING_METHODS:
            continue
        if any(d.endswith(":" + m_) or d.endswith("." + m_) for d in own_defs):
            continue
        errs.append(f"{where[0]}: method :{m_}() is defined by no class in lua/")
    return errs, names


def check_t7():
    """6. lua/catgan/t7.lua (torch.save / torch.load) against its executable twin cat-generator_amd/t7.py: same object tags, same
    class names on disk, same version header."""
    errs = []
    py = open(os.path.join(ROOT, "cat-generator_amd", "t7.py")).read()
    lua = open(os.path.join(ROOT, "lua", "catgan", "t7.lua")).read()
    pat = r"TYPE_NIL, TYPE_NUMBER, TYPE_STRING, TYPE_TABLE, TYPE_TORCH, TYPE_BOOLEAN = ([\d, ]+)"
    a_, b_ = re.search(pat, py), re.search(pat, lua)
    if not a_ or not b_ or [x.strip() for x in a_.group(1).split(",")] != [x.strip() for x in b_.group(1).split(",")]:
        errs.append("lua/catgan/t7.lua: object type tags differ from cat-generator_amd/t7.py")
    for name in ("torch.FloatTensor", "torch.CudaTensor", "torch.FloatStorage", "torch.CudaStorage", "V 1"):
        if name not in lua:
            errs.append(f"lua/catgan/t7.lua does not write '{name}'")
    for name in ("torch.FloatTensor", "torch.CudaTensor", "torch.CudaStorage", "V 1", 'replace("Tensor", "Storage")'):   # t7.py derives the storage names
        if name not in py:
            errs.append(f"cat-generator_amd/t7.py does not write '{name}'")
    for fn in ("int", "long", "double", "string"):
        if not re.search(r"function Writer:%s\(" % fn, lua) or not re.search(r"function Reader:%s\(" % fn, lua):
            errs.append(f"lua/catgan/t7.lua: Writer / Reader lack :{fn}()")
    return errs


def main():
    if "--write-manifest" in sys.argv:
        import json
        refdir = os.environ.get("CATGAN_REFERENCE", "/root/reference")
        json.dump(reference_names(refdir), open(MANIFEST, "w"), indent=1, sort_keys=True)
        print("wrote", MANIFEST)
        return 0
    P = protos()
    errs, ncalls, files = [], 0, []
    for d, _, fs in os.walk(os.path.join(ROOT, "lua")):
        files += [os.path.join(d, f) for f in fs if f.endswith(".lua")]
    defined = set()
    for path in sorted(files):
        rel = os.path.relpath(path, ROOT)
        raw = open(path).read()
        src = strip_lua(raw)
        errs += check_blocks(rel, src)
        errs += check_brackets(rel, raw, src)
        errs += check_strings(rel, raw)
        for m in re.finditer(r"\bC\.(cg_\w+)", src):
            name = m.group(1)
            line = src.count("\n", 0, m.start()) + 1
            if name not in P:
                errs.append(f"{rel}:{line}: {name} is not declared in include/catgan.h"); continue
            k = m.end()
            while k < len(src) and src[k].isspace():
                k += 1
            if k < len(src) and src[k] == "(":
                got, _ = count_args(src, k)
                want = len(P[name][1])
                ncalls += 1
                if got != want:
                    errs.append(f"{rel}:{line}: {name} called with {got} arguments, the header declares {want}")
        for m in re.finditer(r"class\('((?:nn|cudnn)\.\w+)'", raw):
            defined.add(m.group(1))
        for m in re.finditer(r"\bfunction (nn|optim)\.(\w+)", src):
            defined.add(m.group(1) + "." + m.group(2))
    # what models.lua:138-160,196-228,640-711,814-906 / adversarial.lua / train.lua instantiate on the path
    needed = ["nn.Sequential", "nn.Linear", "nn.View", "nn.PReLU", "nn.SpatialUpSamplingNearest", "cudnn.SpatialConvolution",
              "nn.SpatialConvolution", "nn.SpatialBatchNormalization", "nn.Sigmoid", "nn.Copy", "nn.SpatialAveragePooling",
              "nn.SpatialMaxPooling", "nn.SpatialDropout", "nn.Dropout", "nn.Concat", "nn.ConcatTable", "nn.Transpose",
              "nn.AffineTransformMatrixGenerator", "nn.AffineGridGeneratorBHWD", "nn.BilinearSamplerBHWD", "nn.LeakyReLU",
              "nn.SpatialConvolutionUpsample", "cudnn.SpatialConvolutionUpsample", "nn.BCECriterion", "optim.adam", "optim.sgd",
              "optim.adagrad", "optim.ConfusionMatrix"]
    src_all = "".join(open(f).read() for f in files)
    for c in needed:
        if c not in defined and not re.search(r"name\s*=\s*'%s'|pool_class\('%s'" % (re.escape(c), re.escape(c)), src_all):
            errs.append(f"lua/: class or function {c} is not defined")
    errs += check_net_builder()
    errs += check_t7()
    ref_errs, names = check_reference(os.environ.get("CATGAN_REFERENCE", "/root/reference"))
    errs += ref_errs
    for e in errs:
        print(e)
    if names:
        print(f"reference host files {', '.join(names['files'])}: {len(names['requires'])} requires, {len(names['calls'])} namespace calls, "
              f"{len(names['methods'])} method names resolved against lua/ ({len(ref_errs)} unresolved)")
    print(f"{len(files)} Lua files, {ncalls} C-ABI calls checked against {len(P)} prototypes, {len(errs)} problems")
    return 1 if errs else 0


if __name__ == "__main__":
    sys.exit(main())
