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


def main():
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
    for e in errs:
        print(e)
    print(f"{len(files)} Lua files, {ncalls} C-ABI calls checked against {len(P)} prototypes, {len(errs)} problems")
    return 1 if errs else 0


if __name__ == "__main__":
    sys.exit(main())
