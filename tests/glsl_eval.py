"""A small interpreter for the GLSL of GLava's audio utility shaders (TEST INFRASTRUCTURE).

Purpose: the GL twins of the path (gravity_pass.frag, average_pass.frag, smooth.glsl's
smooth_audio) cannot be executed here -- there is no GL context -- and the oracle's C functions for them
(glvo_average_gl, glvo_bars in oracle/glv_oracle.c) are restatements by the same author as the kernels.  This module
is the independent check: it reads the SHADER TEXT from the reference tree (/root/reference/shaders/glava), runs
GLava's own preprocessing on it (#include ":file", #define / function-like macros with ## pasting, #if / #elif /
#else / #endif on integer expressions, GLava's `#expand MACRO COUNT`; glava/glsl_ext.c) and evaluates the resulting
functions with IEEE float32 arithmetic (GLSL highp float), statement by statement.  Nothing here shares code with the
oracle or the kernels; tests/golden/make_glsl_golden.py stores its outputs as golden vectors so the GPU box (which has
no /root/reference) can check against them too.

Supported subset (all these shaders need): float / int / in / highp declarations with initialisers, assignment and
compound assignment, for / if / return, calls of user functions and of log, sin, cos, sqrt, abs, clamp, min, max,
round, int, float, texelFetch(tex, i, 0).r, vec4(...), gl_FragCoord.x, fragment / fragment.r as the output.
"""
from __future__ import annotations

import math
import os
import re

import numpy as np

f32 = np.float32
SHADER_ROOT = "/root/reference/shaders/glava"


# ---------------------------------------------------------------------------------------------- preprocessing
class Macro:
    def __init__(self, params, body):
        self.params, self.body = params, body


def _strip_comments(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


_TOKEN = re.compile(r"[A-Za-z_][A-Za-z_0-9]*|##|\d+\.\d*(?:[eE][-+]?\d+)?[fF]?|\.\d+[fF]?|\d+[fFuU]?|\S")


def _expand(text: str, macros: dict, depth=0) -> str:
    """macro expansion of one logical line (object- and function-like macros, ## pasting)"""
    if depth > 40:
        raise RecursionError("macro recursion")
    out, i = [], 0
    toks = [(m.group(0), m.start(), m.end()) for m in _TOKEN.finditer(text)]
    pos = 0
    k = 0
    res = []
    while k < len(toks):
        tok, s, e = toks[k]
        res.append(text[pos:s])
        pos = e
        if tok in macros and re.match(r"[A-Za-z_]", tok):
            m = macros[tok]
            if m.params is None:
                res.append(_expand(m.body, macros, depth + 1))
            else:
                # function-like: needs '(' next
                if k + 1 < len(toks) and toks[k + 1][0] == "(":
                    depth_p, j, args, cur = 0, k + 1, [], []
                    while j < len(toks):
                        t = toks[j][0]
                        if t == "(":
                            depth_p += 1
                            if depth_p > 1: cur.append(t)
                        elif t == ")":
                            depth_p -= 1
                            if depth_p == 0:
                                break
                            cur.append(t)
                        elif t == "," and depth_p == 1:
                            args.append(" ".join(cur)); cur = []
                        else:
                            cur.append(t)
                        j += 1
                    args.append(" ".join(cur))
                    if len(m.params) == 0: args = []
                    body = m.body
                    # ## pasting first (with raw arguments), then plain substitution with expanded arguments
                    def paste(mm):
                        a, b = mm.group(1), mm.group(2)
                        a = args[m.params.index(a)].strip() if a in m.params else a
                        b = args[m.params.index(b)].strip() if b in m.params else b
                        return a + b
                    body = re.sub(r"([A-Za-z_0-9]+)\s*##\s*([A-Za-z_0-9]+)", paste, body)
                    def sub(mm):
                        w = mm.group(0)
                        return "(" + _expand(args[m.params.index(w)], macros, depth + 1) + ")" if w in m.params and False else \
                               (_expand(args[m.params.index(w)], macros, depth + 1) if w in m.params else w)
                    body = re.sub(r"[A-Za-z_][A-Za-z_0-9]*", sub, body)
                    res.append(_expand(body, macros, depth + 1))
                    pos = toks[j][2]
                    k = j
                else:
                    res.append(tok)
        else:
            res.append(tok)
        k += 1
    res.append(text[pos:])
    out_text = "".join(res)
    # rescan: an object-like macro may have produced the name of a function-like one in front of its argument list
    # (ROUND_FORMULA -> sinusoidal, then `sinusoidal(...)`)
    if depth == 0:
        for _ in range(8):
            again = _expand(out_text, macros, 1)
            if again == out_text: break
            out_text = again
    return out_text


def _eval_pp(expr: str, macros: dict) -> int:
    expr = re.sub(r"defined\s*\(?\s*([A-Za-z_][A-Za-z_0-9]*)\s*\)?", lambda m: "1" if m.group(1) in macros else "0", expr)
    expr = _expand(expr, macros)
    expr = re.sub(r"[A-Za-z_][A-Za-z_0-9]*", "0", expr)          # unknown identifiers are 0, as in C
    expr = expr.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
    return int(eval(expr, {"__builtins__": {}}, {}))


def preprocess(path: str, defines: dict | None = None, root: str = SHADER_ROOT, _macros=None, _seen=None, overrides: dict | None = None,
               _text: str | None = None) -> tuple[str, dict]:
    """returns (expanded source, macro table).  `defines`: the macros GLava itself injects (_AVG_FRAMES, _SMOOTH_FACTOR ...).
    `overrides`: {file name: text} -- a user's copy of a configuration file (~/.config/glava/<name>): GLava includes the installed file first
    ("@name", glsl_ext.c:173-181) and the user's second (":name", :168-172), and re-defines macros textually (:143-157), so the user's wins."""
    macros = _macros if _macros is not None else {k: Macro(None, str(v)) for k, v in (defines or {}).items()}
    src = _strip_comments(_text if _text is not None else open(path).read()).replace("\\\n", " ")
    out = []
    stack = []          # (taking, taken_any)
    def active():
        return all(t for t, _ in stack)
    for raw in src.split("\n"):
        line = raw.strip()
        if line.startswith("#"):
            d = line[1:].strip()
            name = d.split(None, 1)[0] if d else ""
            rest = d[len(name):].strip()
            if name in ("if", "ifdef", "ifndef"):
                if not active(): stack.append((False, True)); continue
                v = _eval_pp(rest, macros) if name == "if" else ((rest.split()[0] in macros) == (name == "ifdef"))
                stack.append((bool(v), bool(v)))
            elif name == "elif":
                t, any_ = stack.pop()
                if not all(x for x, _ in stack): stack.append((False, True)); continue
                v = (not any_) and bool(_eval_pp(rest, macros))
                stack.append((v, any_ or v))
            elif name == "else":
                t, any_ = stack.pop()
                stack.append(((not any_) and all(x for x, _ in stack), True))
            elif name == "endif":
                stack.pop()
            elif not active():
                continue
            elif name == "define":
                m = re.match(r"([A-Za-z_][A-Za-z_0-9]*)(\(([^)]*)\))?\s*(.*)", rest)
                params = None
                if m.group(2) is not None and rest[len(m.group(1))] == "(":
                    params = [p.strip() for p in m.group(3).split(",")] if m.group(3).strip() else []
                macros[m.group(1)] = Macro(params, m.group(4))
            elif name == "undef":
                macros.pop(rest.split()[0], None)
            elif name == "include":
                inc = rest.strip().strip('"')
                if inc.startswith("@") and not (overrides and inc[1:] in overrides):   # the two includes of a configuration file name the same stock file here
                    continue
                p = os.path.join(root, inc[1:]) if inc[0] in ":@" else os.path.join(os.path.dirname(path), inc)
                user = overrides.get(inc[1:]) if overrides and inc.startswith(":") else None
                text, _ = preprocess(p, None, root, macros, overrides=overrides, _text=user)
                out.append(text)
            elif name == "expand":               # GLava's compile-time loop: `#expand MACRO COUNT` (glsl_ext.c)
                mname, cnt = rest.split()
                cnt = _eval_pp(cnt, macros)
                for i in range(cnt):
                    out.append(_expand(f"{mname}({i})", macros) + ";")
            elif name in ("request", "version", "extension", "pragma"):
                continue
            continue
        if active():
            out.append(_expand(raw, macros))
    return "\n".join(out), macros


# ------------------------------------------------------------------------------------------------- evaluation
class _Return(Exception):
    def __init__(self, v): self.v = v


class _Tex:
    """sampler1D bound to a float array; texelFetch(...).r"""
    def __init__(self, data): self.data = np.asarray(data, dtype=np.float32)


class _Texel:
    def __init__(self, r): self.r = f32(r)


def _num(tok: str):
    if re.fullmatch(r"\d+[uU]?", tok):
        return int(tok.rstrip("uU"))
    return f32(float(tok.rstrip("fF")))


def _f(x):
    return x if isinstance(x, (int, np.integer)) and not isinstance(x, bool) else f32(x)


_BUILTINS = {
    "log": lambda x: f32(math.log(float(f32(x)))), "sin": lambda x: f32(math.sin(float(f32(x)))),
    "cos": lambda x: f32(math.cos(float(f32(x)))), "sqrt": lambda x: f32(math.sqrt(float(f32(x)))),
    "abs": lambda x: abs(x), "min": lambda a, b: a if a < b else b, "max": lambda a, b: a if a > b else b,
    "clamp": lambda x, lo, hi: (f32(lo) if x < lo else (f32(hi) if x > hi else f32(x))),
    "round": lambda x: f32(np.round(f32(x))), "int": lambda x: int(x), "float": lambda x: f32(x),
    "vec4": lambda *a: _Texel(a[0]),
}


class Shader:
    """parsed functions of one preprocessed translation unit"""

    def __init__(self, source: str):
        self.src = source
        self.funcs = {}
        for m in re.finditer(r"\b(float|void|int)\s+([A-Za-z_][A-Za-z_0-9]*)\s*\(([^)]*)\)\s*\{", source):
            body_start = m.end()
            depth, i = 1, body_start
            while depth:
                c = source[i]
                depth += (c == "{") - (c == "}")
                i += 1
            params = []
            for p in m.group(3).split(","):
                p = p.strip()
                if p and p != "void":
                    params.append(p.split()[-1])
            self.funcs[m.group(2)] = (params, source[body_start:i - 1])
        self.globals = {}

    # ---- expressions: tokens -> Python objects via a precedence climber
    def _expr(self, text: str, env: dict):
        toks = [t for t in re.findall(r"\d+\.\d*(?:[eE][-+]?\d+)?[fF]?|\.\d+[fF]?|\d+[fFuU]?|[A-Za-z_][A-Za-z_0-9.]*|\.[A-Za-z_][A-Za-z_0-9]*|<=|>=|==|!=|&&|\|\||[-+*/()<>,!]", text)]
        pos = [0]
        def peek(): return toks[pos[0]] if pos[0] < len(toks) else None
        def take():
            t = toks[pos[0]]; pos[0] += 1; return t
        def primary():
            t = take()
            if t == "(":
                v = ternary(); take(); return v
            if t == "-": return -unary_val()
            if t == "+": return unary_val()
            if t == "!": return not unary_val()
            if re.match(r"[\d.]", t): return _num(t)
            name, _, attr = t.partition(".")
            if peek() == "(":
                take()
                args = []
                if peek() != ")":
                    args.append(ternary())
                    while peek() == ",":
                        take(); args.append(ternary())
                take()
                v = self._call(name, args, env)
            else:
                v = env[name] if name in env else self.globals[name]
            if attr:
                v = getattr(v, attr)
            while peek() is not None and peek().startswith(".") and len(peek()) > 1 and not peek()[1].isdigit():
                v = getattr(v, take()[1:])
            return v
        def unary_val(): return primary()
        def arith(a, op, b):
            if isinstance(a, (int, np.integer)) and isinstance(b, (int, np.integer)) and not isinstance(a, bool):
                return {"+": a + b, "-": a - b, "*": a * b, "/": int(a / b) if b else 0}[op]
            a, b = f32(a), f32(b)                      # GLSL: int operands are converted to float
            with np.errstate(all="ignore"):
                return {"+": a + b, "-": a - b, "*": a * b, "/": a / b}[op]
        def mul():
            v = primary()
            while peek() in ("*", "/"):
                op = take(); v = arith(v, op, primary())
            return v
        def add():
            v = mul()
            while peek() in ("+", "-"):
                op = take(); v = arith(v, op, mul())
            return v
        def cmp_():
            v = add()
            while peek() in ("<", ">", "<=", ">=", "==", "!="):
                op = take(); r = add()
                v = {"<": v < r, ">": v > r, "<=": v <= r, ">=": v >= r, "==": v == r, "!=": v != r}[op]
            return v
        def logic():
            v = cmp_()
            while peek() in ("&&", "||"):
                op = take(); r = cmp_()
                v = (v and r) if op == "&&" else (v or r)
            return v
        def ternary(): return logic()
        return ternary()

    def _call(self, name, args, env):
        if name == "texelFetch":
            tex, i = args[0], int(args[1])
            d = tex.data
            return _Texel(d[i] if 0 <= i < d.size else 0.0)      # out-of-range texelFetch is undefined in GL; 0 here
        if name in self.funcs:
            return self.call(name, *args)
        return _BUILTINS[name](*args)

    # ---- statements
    def _split_top(self, text, sep):
        parts, depth, cur = [], 0, []
        for c in text:
            if c in "([{": depth += 1
            if c in ")]}": depth -= 1
            if c == sep and depth == 0:
                parts.append("".join(cur)); cur = []
            else:
                cur.append(c)
        parts.append("".join(cur))
        return parts

    def _simple(self, stmt: str, env: dict):
        stmt = stmt.strip()
        if not stmt: return
        if stmt.startswith("return"):
            raise _Return(self._expr(stmt[6:], env) if stmt[6:].strip() else None)
        m = re.match(r"(?:(?:highp|mediump|lowp|in|const)\s+)*(float|int)\s+(.*)", stmt, flags=re.S)
        if m:                                         # declaration list
            is_int = m.group(1) == "int"
            for d in self._split_top(m.group(2), ","):
                name, _, init = d.partition("=")
                v = self._expr(init, env) if init.strip() else (0 if is_int else f32(0))
                env[name.strip()] = int(v) if is_int else f32(v)
            return
        m = re.match(r"([A-Za-z_][A-Za-z_0-9.]*)\s*(\+=|-=|\*=|/=|=)(?!=)\s*(.*)", stmt, flags=re.S)
        if m:
            name, op, rhs = m.group(1), m.group(2), self._expr(m.group(3), env)
            base = name.split(".")[0]
            if base == "fragment":
                env["fragment"] = f32(rhs.r if isinstance(rhs, _Texel) else rhs)
                return
            cur = env.get(name)
            if op != "=":
                a, b = f32(cur), f32(rhs)
                with np.errstate(all="ignore"):
                    rhs = {"+=": a + b, "-=": a - b, "*=": a * b, "/=": a / b}[op]
            env[name] = int(rhs) if isinstance(cur, int) and not isinstance(cur, bool) and op == "=" and isinstance(rhs, int) else f32(rhs)
            return
        self._expr(stmt, env)

    def _block(self, text: str, env: dict):
        i, n = 0, len(text)
        while i < n:
            while i < n and text[i] in " \t\r\n;": i += 1
            if i >= n: break
            m = re.match(r"(for|if)\s*\(", text[i:])
            if m:
                j = i + m.end(); depth = 1
                while depth:
                    depth += (text[j] == "(") - (text[j] == ")"); j += 1
                head = text[i + m.end():j - 1]
                while text[j] in " \t\r\n": j += 1
                if text[j] == "{":
                    k, depth = j + 1, 1
                    while depth:
                        depth += (text[k] == "{") - (text[k] == "}"); k += 1
                    body, nxt = text[j + 1:k - 1], k
                else:
                    k = text.index(";", j); body, nxt = text[j:k + 1], k + 1
                if m.group(1) == "if":
                    if self._expr(head, env): self._block(body, env)
                    # optional else
                    me = re.match(r"\s*else\s*", text[nxt:])
                    if me:
                        jj = nxt + me.end()
                        if text[jj] == "{":
                            kk, depth = jj + 1, 1
                            while depth:
                                depth += (text[kk] == "{") - (text[kk] == "}"); kk += 1
                            if not self._expr(head, env): self._block(text[jj + 1:kk - 1], env)
                            nxt = kk
                        else:
                            kk = text.index(";", jj)
                            if not self._expr(head, env): self._block(text[jj:kk + 1], env)
                            nxt = kk + 1
                else:
                    init, cond, step = self._split_top(head, ";")
                    self._simple(init, env)
                    guard = 0
                    while self._expr(cond, env):
                        self._block(body, env)
                        self._simple(step, env)
                        guard += 1
                        if guard > 1 << 22: raise RuntimeError("runaway loop")
                i = nxt
                continue
            k = i; depth = 0
            while k < n and not (text[k] == ";" and depth == 0):
                depth += (text[k] in "([") - (text[k] in ")]"); k += 1
            self._simple(text[i:k], env)
            i = k + 1

    def call(self, name, *args):
        params, body = self.funcs[name]
        env = dict(zip(params, [a if isinstance(a, (_Tex, int)) and not isinstance(a, bool) else (a if isinstance(a, _Tex) else f32(a)) for a in args]))
        for p, a in zip(params, args):
            if isinstance(a, (int, np.integer)) and not isinstance(a, bool): env[p] = int(a)
        try:
            self._block(body, env)
        except _Return as r:
            return r.v
        return env.get("fragment")


# ------------------------------------------------------------------------------------------ the shaders of the path
def load(relpath: str, defines: dict, overrides: dict | None = None) -> Shader:
    src, _ = preprocess(os.path.join(SHADER_ROOT, relpath), defines, overrides=overrides)
    return Shader(src)


def smooth_audio_bars(tex_row, bars: int, smooth_factor=0.025, pre_smoothed=0, user_parameters: str | None = None, phase=0.0) -> np.ndarray:
    """smooth_audio(tex, n, (k + phase) / bars) of shaders/glava/util/smooth.glsl for k = 0 .. bars-1, evaluated from the shader text.
    user_parameters: the text of a user's smooth_parameters.glsl (`#define SAMPLE_MODE maximum` ...), included after the stock one"""
    sh = load("util/smooth.glsl", {"_SMOOTH_FACTOR": repr(float(smooth_factor)), "_PRE_SMOOTHED_AUDIO": pre_smoothed},
              {"smooth_parameters.glsl": user_parameters} if user_parameters is not None else None)
    tex = _Tex(tex_row)
    n = int(tex.data.size)
    return np.array([sh.call("smooth_audio", tex, n, (f32(k) + f32(phase)) / f32(bars) if phase else f32(k) / f32(bars)) for k in range(bars)], np.float32)


def average_pass(frames_newest_first, avg_window=True) -> np.ndarray:
    """shaders/glava/util/average_pass.frag: one output texel per x from _AVG_FRAMES textures t0 (newest) .. t{F-1}"""
    F = len(frames_newest_first)
    sh = load("util/average_pass.frag", {"_AVG_FRAMES": F, "_AVG_WINDOW": int(bool(avg_window))})
    texs = [_Tex(f) for f in frames_newest_first]
    for i, t in enumerate(texs): sh.globals[f"t{i}"] = t
    n = texs[0].data.size
    out = np.empty(n, np.float32)

    class Frag: pass
    for x in range(n):
        fc = Frag(); fc.x = f32(x) + f32(0.5)                   # pixel centres
        sh.globals["gl_FragCoord"] = fc
        out[x] = sh.call("main")
    return out


def gravity_pass(store_row, diff) -> np.ndarray:
    """shaders/glava/util/gravity_pass.frag: fragment.r = texel - diff"""
    sh = load("util/gravity_pass.frag", {})
    sh.globals["tex"] = _Tex(store_row); sh.globals["diff"] = f32(diff)
    n = sh.globals["tex"].data.size
    out = np.empty(n, np.float32)

    class Frag: pass
    for x in range(n):
        fc = Frag(); fc.x = f32(x) + f32(0.5)
        sh.globals["gl_FragCoord"] = fc
        out[x] = sh.call("main")
    return out
