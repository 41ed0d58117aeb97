"""JSFX/EEL2 tokeniser and expression parser for the zajit translator.

The grammar accepted here is the one the reference's AOT compiler defines -- that compiler, not REAPER,
is what fixes the product semantics of the hot path:
  tokens ............ dsp_jsfx_aot.py:81-250   (dotted identifiers, '#'/'$' names, quoted literals)
  binding powers .... dsp_jsfx_aot.py:370-389  ('|' with '||', '&' with '==', '^' tightest, left-assoc)
  statements ........ dsp_jsfx_aot.py:455-577  (if/else, while(c) body, function name(p) local() instance())
  expressions ....... dsp_jsfx_aot.py:593-828  (newline-led infix continuation, `c ? a` with implicit else 0,
                                                loop(n, body...), calls, a[b], paren sequences)
This is an independent implementation: a token array with an index cursor and a binding-power climb.
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import List, Optional

# ----------------------------------------------------------------------------------------------
# AST
# ----------------------------------------------------------------------------------------------


class Node:
    __slots__ = ("line", "col")


def _mk(name, fields):
    def __init__(self, *args, line=0, col=0):
        assert len(args) == len(fields), (name, args)
        for f, a in zip(fields, args):
            setattr(self, f, a)
        self.line, self.col = line, col

    def __repr__(self):
        return f"{name}(" + ", ".join(f"{f}={getattr(self, f)!r}" for f in fields) + ")"

    return type(name, (Node,), {"__slots__": tuple(fields), "__init__": __init__, "__repr__": __repr__,
                                "_fields": tuple(fields)})


Num = _mk("Num", ["value"])
Str = _mk("Str", ["value"])
Var = _mk("Var", ["name"])
Index = _mk("Index", ["base", "index"])
Unary = _mk("Unary", ["op", "a"])
Binary = _mk("Binary", ["op", "l", "r"])
Assign = _mk("Assign", ["op", "target", "value"])
Call = _mk("Call", ["fn", "args"])
Loop = _mk("Loop", ["count", "body"])
Cond = _mk("Cond", ["cond", "then", "els"])          # c ? a : b
Seq = _mk("Seq", ["items"])
If = _mk("If", ["cond", "then", "els"])              # if (c) a else b   (statement, value 0)
While = _mk("While", ["cond", "body"])
FuncDef = _mk("FuncDef", ["name", "params", "locals", "instances", "body"])


def children(n: Node):
    """Direct child nodes in evaluation-independent order (for generic walks)."""
    for f in n._fields:
        v = getattr(n, f)
        if isinstance(v, Node):
            yield v
        elif isinstance(v, list):
            for it in v:
                if isinstance(it, Node):
                    yield it


def rebuild(n: Node, fn):
    """Copy of n with every child node replaced by fn(child)."""
    vals = []
    for f in n._fields:
        v = getattr(n, f)
        if isinstance(v, Node):
            v = fn(v)
        elif isinstance(v, list) and v and isinstance(v[0], Node):
            v = [fn(it) for it in v]
        vals.append(v)
    return type(n)(*vals, line=n.line, col=n.col)


# ----------------------------------------------------------------------------------------------
# Tokens
# ----------------------------------------------------------------------------------------------

@dataclass(frozen=True)
class Tok:
    kind: str      # eof eol num id kw op punc semi str
    text: str
    line: int
    col: int


_TWO_CHAR_OPS = {"==", "!=", "<=", ">=", "+=", "-=", "*=", "/=", "%=", "^=", "|=", "&=", "~=", "&&", "||", "<<", ">>"}
_NUM_RE = re.compile(r"[0-9]+(\.[0-9]*)?([eE][+-]?[0-9]+)?|\.[0-9]+([eE][+-]?[0-9]+)?")
_ID_RE = re.compile(r"[#$A-Za-z_][#$A-Za-z0-9_]*(?:\.[#$A-Za-z_][#$A-Za-z0-9_]*)*")
_ESCAPES = {"n": "\n", "r": "\r", "t": "\t", "\\": "\\", "0": "\0"}


class JsfxSyntaxError(SyntaxError):
    pass


def tokenize(src: str, base_line: int = 1) -> List[Tok]:
    toks: List[Tok] = []
    i, n, line, col = 0, len(src), base_line, 1

    def err(msg):
        raise JsfxSyntaxError(f"{msg} at {line}:{col}")

    while i < n:
        c = src[i]
        if c in " \t\r":
            i += 1; col += 1
            continue
        if c == "\n":
            toks.append(Tok("eol", "\n", line, col))
            i += 1; line += 1; col = 1
            continue
        if c == "/" and i + 1 < n and src[i + 1] == "/":
            while i < n and src[i] != "\n":
                i += 1
            continue
        if c == "/" and i + 1 < n and src[i + 1] == "*":
            j = src.find("*/", i + 2)
            if j < 0:
                raise JsfxSyntaxError("Unterminated /* comment */")
            seg = src[i:j + 2]
            nl = seg.count("\n")
            if nl:
                line += nl
                col = len(seg) - seg.rfind("\n")
            else:
                col += len(seg)
            i = j + 2
            continue
        if src[i:i + 2] in _TWO_CHAR_OPS:
            toks.append(Tok("op", src[i:i + 2], line, col))
            i += 2; col += 2
            continue
        if c.isdigit() or (c == "." and i + 1 < n and src[i + 1].isdigit()):
            m = _NUM_RE.match(src, i)
            toks.append(Tok("num", m.group(0), line, col))
            col += m.end() - i; i = m.end()
            continue
        if c.isalpha() or c in "_$#":
            m = _ID_RE.match(src, i)
            t = m.group(0)
            toks.append(Tok("kw" if t in ("if", "else", "while") else "id", t, line, col))
            col += m.end() - i; i = m.end()
            continue
        if c in "\"'":
            q, j, out = c, i + 1, []
            while True:
                if j >= n:
                    err("Unterminated string literal")
                ch = src[j]
                if ch in "\n\r":
                    err("Newline in string literal")
                if ch == q:
                    j += 1
                    break
                if ch == "\\":
                    if j + 1 >= n:
                        err("Unterminated string escape")
                    e = src[j + 1]
                    j += 2
                    if e in "xX" and re.fullmatch(r"[0-9A-Fa-f]{2}", src[j:j + 2] or ""):
                        out.append(chr(int(src[j:j + 2], 16)))
                        j += 2
                    elif e == q:
                        out.append(q)
                    else:
                        out.append(_ESCAPES.get(e, e))
                    continue
                out.append(ch)
                j += 1
            toks.append(Tok("str", "".join(out), line, col))
            col += j - i; i = j
            continue
        if c == ";":
            toks.append(Tok("semi", c, line, col))
        elif c in "()[]{},":
            toks.append(Tok("punc", c, line, col))
        elif c in "+-*/=<>&|!?:%~^":
            toks.append(Tok("op", c, line, col))
        else:
            err(f"Unexpected character {c!r}")
        i += 1; col += 1
    toks.append(Tok("eof", "", line, col))
    return toks


# ----------------------------------------------------------------------------------------------
# Parser
# ----------------------------------------------------------------------------------------------

_ASSIGN_OPS = {"=", "+=", "-=", "*=", "/=", "%=", "^=", "|=", "&=", "~="}
_BP = {"||": 3, "|": 3, "&&": 4, "==": 5, "!=": 5, "&": 5, "<": 6, "<=": 6, ">": 6, ">=": 6, "<<": 6, ">>": 6,
       "+": 7, "-": 7, "*": 8, "/": 8, "%": 8, "^": 9}
_BP.update({op: 1 for op in _ASSIGN_OPS})
_COND_BP = 2


class Parser:
    def __init__(self, src: str, base_line: int = 1):
        self.lines = src.splitlines()
        self.base_line = base_line
        self.t = tokenize(src, base_line)
        self.p = 0

    # -- cursor helpers
    @property
    def cur(self) -> Tok:
        return self.t[self.p]

    @property
    def nxt(self) -> Tok:
        return self.t[min(self.p + 1, len(self.t) - 1)]

    def adv(self) -> Tok:
        tk = self.t[self.p]
        if self.p < len(self.t) - 1:
            self.p += 1
        return tk

    def at(self, kind, text=None) -> bool:
        c = self.cur
        return c.kind == kind and (text is None or c.text == text)

    def fail(self, msg):
        c = self.cur
        rel = c.line - self.base_line
        src_line = self.lines[rel] if 0 <= rel < len(self.lines) else ""
        caret = " " * max(0, min(c.col, len(src_line) + 1) - 1) + "^" if src_line else ""
        raise JsfxSyntaxError(f"{msg} at {c.line}:{c.col}" + (f"\n{src_line}\n{caret}" if src_line else ""))

    def eat(self, kind, text=None) -> Tok:
        if self.cur.kind != kind:
            self.fail(f"Expected {kind}, got {self.cur.kind} {self.cur.text!r}")
        if text is not None and self.cur.text != text:
            self.fail(f"Expected {text!r}, got {self.cur.text!r}")
        return self.adv()

    def skip_seps(self):
        while self.cur.kind in ("eol", "semi"):
            self.adv()

    def skip_eol(self):
        while self.cur.kind == "eol":
            self.adv()

    # -- program / statements
    def parse_program(self) -> List[Node]:
        out = []
        self.skip_seps()
        while not self.at("eof"):
            out.append(self.statement(top=True))
            self.skip_seps()
        return out

    def statement(self, top=False) -> Node:
        if self.at("kw", "if"):
            return self.if_stmt()
        if self.at("kw", "while"):
            return self.while_stmt()
        if top and self.at("id", "function"):
            return self.function_def()
        return self.expr(0)

    def if_stmt(self) -> Node:
        kw = self.eat("kw", "if")
        self.eat("punc", "(")
        cond = self.expr(0)
        self.eat("punc", ")")
        self.skip_seps()
        then = self.expr(0)
        self.skip_seps()
        els = None
        if self.at("kw", "else"):
            self.adv()
            self.skip_seps()
            els = self.expr(0)
            self.skip_seps()
        return If(cond, then, els, line=kw.line, col=kw.col)

    def while_stmt(self) -> Node:
        kw = self.eat("kw", "while")
        self.eat("punc", "(")
        cond = self.expr(0)
        self.eat("punc", ")")
        self.skip_seps()
        body = self.expr(0)
        return While(cond, body, line=kw.line, col=kw.col)

    def _name_list(self, what) -> List[str]:
        names = []
        self.eat("punc", "(")
        self.skip_seps()
        while not self.at("punc", ")"):
            if self.cur.kind != "id":
                self.fail(f"Expected {what} name")
            names.append(self.adv().text)
            self.skip_seps()
            if self.at("punc", ","):
                self.adv()
                self.skip_seps()
                continue
            if self.cur.kind == "id":
                continue
            break
        self.skip_seps()
        self.eat("punc", ")")
        return names

    def function_def(self) -> Node:
        kw = self.eat("id", "function")
        if self.cur.kind != "id":
            self.fail("Expected function name after 'function'")
        name = self.adv().text
        params = self._name_list("parameter")
        locals_, instances = [], []
        self.skip_seps()
        while self.cur.kind == "id" and self.cur.text in ("local", "instance", "global"):
            q = self.adv().text
            names = self._name_list(f"{q} variable")
            if q == "local":
                locals_ += names
            elif q == "instance":
                instances += names
            self.skip_seps()
        if not self.at("punc", "("):
            self.fail("Expected '(' to start function body")
        body = self.primary()
        self.skip_seps()
        if self.at("semi"):
            self.adv()
        return FuncDef(name, params, locals_, instances, body, line=kw.line, col=kw.col)

    # -- expressions
    def _continues(self, tok: Tok, min_bp: int) -> bool:
        if tok.kind != "op":
            return False
        if tok.text == "?":
            return _COND_BP >= min_bp
        if tok.text in (":", "+", "-", "!"):
            return False
        bp = _BP.get(tok.text)
        return bp is not None and bp >= min_bp

    def expr(self, min_bp: int) -> Node:
        lhs = self.prefix()
        while True:
            while self.cur.kind == "eol" and (self.nxt.kind == "eol" or self._continues(self.nxt, min_bp)):
                self.adv()
            if self.cur.kind != "op" or self.cur.text in ("?", ":"):
                break
            op = self.cur.text
            bp = _BP.get(op)
            if bp is None or bp < min_bp:
                break
            self.adv()
            if op in _ASSIGN_OPS:
                rhs = self.expr(bp)
                ok = isinstance(lhs, (Var, Index)) or (isinstance(lhs, Call) and lhs.fn in ("slider", "spl") and len(lhs.args) == 1)
                if not ok:
                    self.fail("Assignment target must be a variable, index, or slider()/spl() reference")
                lhs = Assign(op, lhs, rhs, line=lhs.line, col=lhs.col)
            else:
                rhs = self.expr(bp + 1)
                lhs = Binary(op, lhs, rhs, line=lhs.line, col=lhs.col)
        while self.cur.kind == "eol" and (self.nxt.kind == "eol" or (self.nxt.kind == "op" and self.nxt.text == "?")):
            self.adv()
        if self.at("op", "?") and _COND_BP >= min_bp:
            q = self.adv()
            self.skip_seps()
            then = self.expr(0)
            self.skip_seps()
            if self.at("op", ":"):
                self.adv()
                self.skip_seps()
                els = self.expr(0)
            else:
                els = Num(0.0, line=q.line, col=q.col)
            lhs = Cond(lhs, then, els, line=q.line, col=q.col)
        return lhs

    def prefix(self) -> Node:
        self.skip_eol()
        if self.cur.kind == "op" and self.cur.text in ("+", "-", "!"):
            tk = self.adv()
            return Unary(tk.text, self.prefix(), line=tk.line, col=tk.col)
        return self.postfix()

    def seq_item(self) -> Node:
        return self.statement(top=False)

    def postfix(self) -> Node:
        node = self.primary()
        while True:
            if self.at("punc", "("):
                lp = self.adv()
                if not isinstance(node, Var):
                    self.fail("Can only call a named function")
                fn = node.name
                if fn == "loop":
                    self.skip_seps()
                    count = self.expr(0)
                    self.skip_seps()
                    if self.at("punc", ","):
                        self.adv()
                    self.skip_seps()
                    items = []
                    while not self.at("punc", ")"):
                        items.append(self.seq_item())
                        self.skip_seps()
                    self.adv()
                    if not items:
                        body = Num(0.0, line=lp.line, col=lp.col)
                    elif len(items) == 1:
                        body = items[0]
                    else:
                        body = Seq(items, line=lp.line, col=lp.col)
                    node = Loop(count, body, line=lp.line, col=lp.col)
                    continue
                args = []
                self.skip_seps()
                if not self.at("punc", ")"):
                    while True:
                        self.skip_seps()
                        args.append(self.expr(0))
                        self.skip_seps()
                        if self.at("punc", ","):
                            self.adv()
                            continue
                        break
                self.skip_seps()
                self.eat("punc", ")")
                node = Call(fn, args, line=lp.line, col=lp.col)
                continue
            if self.at("punc", "["):
                lb = self.adv()
                self.skip_seps()
                if self.at("punc", "]"):
                    idx = Num(0.0, line=lb.line, col=lb.col)
                else:
                    idx = self.expr(0)
                    self.skip_seps()
                self.eat("punc", "]")
                node = Index(node, idx, line=lb.line, col=lb.col)
                continue
            return node

    def primary(self) -> Node:
        c = self.cur
        if c.kind == "num":
            self.adv()
            return Num(float(c.text), line=c.line, col=c.col)
        if c.kind == "str":
            self.adv()
            return Str(c.text, line=c.line, col=c.col)
        if c.kind == "id":
            self.adv()
            return Var(c.text, line=c.line, col=c.col)
        if self.at("punc", "("):
            self.adv()
            self.skip_seps()
            if self.at("punc", ")"):
                self.adv()
                return Seq([], line=c.line, col=c.col)
            items = [self.seq_item()]
            if self.at("punc", ")"):
                self.adv()
                return items[0]
            while True:
                self.skip_seps()
                if self.at("punc", ")"):
                    self.adv()
                    break
                items.append(self.seq_item())
            return Seq(items, line=c.line, col=c.col)
        self.fail("Expected number, identifier, or '('")


def parse_section(code: str, base_line: int = 1) -> List[Node]:
    return Parser(code, base_line).parse_program()
