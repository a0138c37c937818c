#!/usr/bin/env python3
"""Dev-time derivation of the 64-point forward lifting DCT network and the
coefficient scan permutation used by FFV2.

This script is NOT part of the product or of the test-suite.  It needs the
read-only reference checkout (default /root/reference) and is run once, in the
build container, to (re)generate the files under ``tools/ir/``:

  fdct64_ir.json    the 1-D network as a flat register-machine op list (our IR)
  scan_lut.json     coding-order -> raster-offset permutation (4096 entries)

How it works (no reference text is copied into the repo):

* The macro block that spells the network (reference libavcodec/ffv2.c:313-4001,
  ``OD_*`` lifting macros) and the body of ``od_bin_fdct64`` (ffv2.c:4678-4812)
  are excerpted into a temp dir and expanded by ``gcc -E -P`` (pure text
  pre-processing, no headers).  The expanded straight-line statements are then
  *symbolically executed*: every C statement is parsed with Python's ``ast``
  (the statement grammar used there -- +, -, *, >>, <, names, int literals -- is
  a subset of Python's) with C block scoping for the ``do { dctcoef x; ... }
  while (0)`` temporaries, and lowered to six op kinds on virtual registers:

      SUB d a b        d = a - b
      ADD d a b        d = a + b
      RSH1 d a         d = (a + (a < 0)) >> 1          (OD_RSHIFT1, ffv2.c:313)
      MLA d a K R S    d = d + ((a*K + R) >> S)
      MLS d a K R S    d = d - ((a*K + R) >> S)
      NEG d a          d = -a

* The scan tables (reference libavcodec/zigzags.h, five ``layout_freq_*``
  initialisers, walked by ``raster_to_coding`` ffv2.c:62-79) are parsed as data
  into one permutation; the 4x4 table's missing 16th entry is resolved the way
  the ELF build resolves it (zero padding -> {0,0}).

The same expanded statements are also executed *numerically* on numpy int64
vectors to write ``tests/golden/fdct64_vectors.npz`` (inputs + outputs): those
vectors come from the reference's own statement text, not from our IR, so they
pin both the IR and every consumer of it.
"""
import argparse
import ast
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


# --------------------------------------------------------------------------
# 1. excerpt + preprocess
# --------------------------------------------------------------------------
def expanded_statements(ref, func="od_bin_fdct64"):
    src = open(os.path.join(ref, "libavcodec/ffv2.c")).read().split("\n")
    start = next(i for i, l in enumerate(src) if l.startswith("#define OD_RSHIFT1"))
    end = next(i for i, l in enumerate(src) if l.startswith("static void od_bin_fdct4("))
    f0 = next(i for i, l in enumerate(src) if l.startswith("static void %s(" % func))
    f1 = next(i for i in range(f0, len(src)) if src[i] == "}")
    text = ["#define dctcoef int", "#define OD_DCT_OVERFLOW_CHECK(a,b,c,d)"]
    text += src[start:end] + src[f0:f1 + 1]
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "excerpt.c")
        open(p, "w").write("\n".join(text) + "\n")
        out = subprocess.run(["gcc", "-E", "-P", p], check=True,
                             capture_output=True, text=True).stdout
    body = out[out.index("{") + 1: out.rindex("}")]
    return body


def tokenize(body):
    """Yield ('open',), ('close',), ('decl', name, expr|None), ('stmt', text)."""
    body = re.sub(r"\s+", " ", body)
    pos = 0
    n = len(body)
    while pos < n:
        if body[pos] == " ":
            pos += 1
            continue
        m = re.match(r"do \{", body[pos:])
        if m:
            yield ("open",)
            pos += m.end()
            continue
        m = re.match(r"\} while \(0\) ?;", body[pos:])
        if m:
            yield ("close",)
            pos += m.end()
            continue
        if body[pos] == ";":          # empty statement left by the empty overflow macro
            pos += 1
            continue
        semi = body.index(";", pos)
        st = body[pos:semi].strip()
        pos = semi + 1
        m = re.match(r"int (\w+)(?: = (.*))?$", st)
        if m:
            yield ("decl", m.group(1), m.group(2))
        else:
            yield ("stmt", st)


# --------------------------------------------------------------------------
# 2. symbolic execution -> IR
# --------------------------------------------------------------------------
class Lowerer:
    def __init__(self):
        self.scopes = [{}]
        self.nreg = 0
        self.ops = []
        self.inputs = {}     # natural input index -> reg
        self.outputs = {}    # natural output index -> reg

    def new(self):
        r = self.nreg
        self.nreg += 1
        return r

    def declare(self, name):
        r = self.new()
        self.scopes[-1][name] = r
        return r

    def lookup(self, name):
        for s in reversed(self.scopes):
            if name in s:
                return s[name]
        raise KeyError(name)

    # expression helpers ---------------------------------------------------
    def is_name(self, e):
        return isinstance(e, ast.Name)

    def const(self, e):
        if isinstance(e, ast.Constant) and isinstance(e.value, int):
            return e.value
        raise ValueError(ast.dump(e))

    def match_rsh1(self, e):
        # ((a + (a < 0)) >> 1)
        if (isinstance(e, ast.BinOp) and isinstance(e.op, ast.RShift)
                and isinstance(e.right, ast.Constant) and e.right.value == 1
                and isinstance(e.left, ast.BinOp) and isinstance(e.left.op, ast.Add)
                and self.is_name(e.left.left)
                and isinstance(e.left.right, ast.Compare)
                and self.is_name(e.left.right.left)
                and e.left.right.left.id == e.left.left.id
                and isinstance(e.left.right.ops[0], ast.Lt)
                and self.const(e.left.right.comparators[0]) == 0):
            return e.left.left.id
        return None

    def match_mulshift(self, e):
        # (a*K + R) >> S
        if (isinstance(e, ast.BinOp) and isinstance(e.op, ast.RShift)
                and isinstance(e.left, ast.BinOp) and isinstance(e.left.op, ast.Add)
                and isinstance(e.left.left, ast.BinOp)
                and isinstance(e.left.left.op, ast.Mult)
                and self.is_name(e.left.left.left)):
            return (e.left.left.left.id, self.const(e.left.left.right),
                    self.const(e.left.right), self.const(e.right))
        return None

    def atom(self, e):
        """A name, or an OD_RSHIFT1 of a name (lowered through a fresh temp)."""
        if self.is_name(e):
            return self.lookup(e.id)
        r = self.match_rsh1(e)
        if r is not None:
            t = self.new()
            self.ops.append(["RSH1", t, self.lookup(r)])
            return t
        return None

    def statement(self, text):
        m = re.match(r"(?:y\[(\d+)\]|x\[(\d+)\*xstride\]) = (\w+)$", text)
        if m:
            self.outputs[int(m.group(1) or m.group(2))] = self.lookup(m.group(3))
            return
        node = ast.parse(text).body[0]
        if isinstance(node, ast.AugAssign):
            d = self.lookup(node.target.id)
            sign = {ast.Add: +1, ast.Sub: -1}[type(node.op)]
            v = node.value
            if self.is_name(v):
                self.ops.append(["ADD" if sign > 0 else "SUB", d, d, self.lookup(v.id)])
                return
            ms = self.match_mulshift(v)
            if ms:
                a, K, R, S = ms
                self.ops.append(["MLA" if sign > 0 else "MLS", d, self.lookup(a), K, R, S])
                return
            r = self.match_rsh1(v)
            if r is not None:
                t = self.new()
                self.ops.append(["RSH1", t, self.lookup(r)])
                self.ops.append(["ADD" if sign > 0 else "SUB", d, d, t])
                return
            raise ValueError("unhandled augassign: " + text)
        if isinstance(node, ast.Assign):
            d = self.lookup(node.targets[0].id)
            v = node.value
            r = self.match_rsh1(v)
            if r is not None:
                self.ops.append(["RSH1", d, self.lookup(r)])
                return
            if isinstance(v, ast.BinOp) and isinstance(v.op, (ast.Add, ast.Sub)):
                a, b = self.atom(v.left), self.atom(v.right)
                if a is not None and b is not None:
                    self.ops.append(["ADD" if isinstance(v.op, ast.Add) else "SUB", d, a, b])
                    return
            if isinstance(v, ast.UnaryOp) and isinstance(v.op, ast.USub) and self.is_name(v.operand):
                self.ops.append(["NEG", d, self.lookup(v.operand.id)])
                return
            if self.is_name(v):
                self.ops.append(["MOV", d, self.lookup(v.id)])
                return
            raise ValueError("unhandled assign: " + text)
        raise ValueError("unhandled statement: " + text)

    def run(self, body):
        for tok in tokenize(body):
            if tok[0] == "open":
                self.scopes.append({})
            elif tok[0] == "close":
                self.scopes.pop()
            elif tok[0] == "decl":
                r = self.declare(tok[1])
                if tok[2] is not None:
                    m = re.match(r"(?:x\[(\d+)\*xstride\]|y\[(\d+)\])$", tok[2])
                    if not m:
                        raise ValueError("unhandled initialiser: " + tok[2])
                    self.inputs[int(m.group(1) or m.group(2))] = r
            else:
                self.statement(tok[1])
        assert sorted(self.inputs) == list(range(64))
        assert sorted(self.outputs) == list(range(64))


def compact(low):
    """Renumber registers: the 64 inputs become r0..r63 in natural order."""
    remap = {low.inputs[k]: k for k in range(64)}
    nxt = 64
    ops = []
    for op in low.ops:
        regs = [1, 2] if op[0] in ("RSH1", "MLA", "MLS", "NEG", "MOV") else [1, 2, 3]
        op = list(op)
        for i in regs:
            if op[i] not in remap:
                remap[op[i]] = nxt
                nxt += 1
            op[i] = remap[op[i]]
        ops.append(op)
    outs = [remap[low.outputs[k]] for k in range(64)]
    return {"n_in": 64, "n_regs": nxt, "ops": ops, "out_regs": outs}


# --------------------------------------------------------------------------
# 3. numeric execution of the *statement text* (golden vectors; independent of IR)
# --------------------------------------------------------------------------
def numeric_golden(body, xs):
    """xs: (N,64) int64.  Execute the expanded statements with numpy vectors."""
    scopes = [{}]
    uid = [0]
    env = {}
    ys = np.zeros_like(xs)

    def resolve(name):
        for s in reversed(scopes):
            if name in s:
                return s[name]
        raise KeyError(name)

    def rewrite(text):
        return re.sub(r"\b([A-Za-z_]\w*)\b", lambda m: resolve(m.group(1)), text)

    for tok in tokenize(body):
        if tok[0] == "open":
            scopes.append({})
        elif tok[0] == "close":
            scopes.pop()
        elif tok[0] == "decl":
            u = "v%d" % uid[0]
            uid[0] += 1
            scopes[-1][tok[1]] = u
            if tok[2] is not None:
                mm = re.match(r"(?:x\[(\d+)\*xstride\]|y\[(\d+)\])$", tok[2])
                k = int(mm.group(1) or mm.group(2))
                env[u] = xs[:, k].copy()
            else:
                env[u] = np.zeros(xs.shape[0], dtype=np.int64)
        else:
            m = re.match(r"(?:y\[(\d+)\]|x\[(\d+)\*xstride\]) = (\w+)$", tok[1])
            if m:
                ys[:, int(m.group(1) or m.group(2))] = env[resolve(m.group(3))]
                continue
            # (a < 0) must be 0/1 ints, >> is arithmetic on int64: numpy does both.
            py = rewrite(tok[1])
            py = re.sub(r"\((\w+)\) < 0", r"((\1) < 0).astype(np.int64)", py)
            exec(py, {"np": np}, env)
    lim = 2 ** 31
    return ys


# --------------------------------------------------------------------------
# 4. scan tables
# --------------------------------------------------------------------------
def scan_lut(ref):
    txt = open(os.path.join(ref, "libavcodec/zigzags.h")).read()
    lut = []
    for size, want in ((4, 16), (8, 48), (16, 192), (32, 768), (64, 3072)):
        m = re.search(r"layout_freq_%dx%d = \{(.*?)\n\};" % (size, size), txt, re.S)
        blk = m.group(1)
        zlen = int(re.match(r"\s*(\d+),", blk).group(1))
        assert zlen == want
        inner = blk[blk.index("{", blk.index("{") + 1 if size == 4 else 0):]
        pairs = re.findall(r"\{\s*(\d+),\s*(\d+)\s*\}", blk)
        pairs = [(int(a), int(b)) for a, b in pairs]
        if size == 4:
            # only 15 initialisers for zigzag_len 16: the 16th reads the zero bytes
            # after the object -> {0,0}.
            assert len(pairs) == 15
            pairs.append((0, 0))
        assert len(pairs) == zlen, (size, len(pairs))
        lut += [y * 64 + x for x, y in pairs]
    assert sorted(lut) == list(range(4096))
    return lut


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()

    body = expanded_statements(args.ref)
    low = Lowerer()
    low.run(body)
    ir = compact(low)
    kinds = {}
    for op in ir["ops"]:
        kinds[op[0]] = kinds.get(op[0], 0) + 1
    print("ops:", len(ir["ops"]), kinds, "regs:", ir["n_regs"])

    gen = os.path.join(ROOT, "tools", "ir")
    os.makedirs(gen, exist_ok=True)
    json.dump(ir, open(os.path.join(gen, "fdct64_ir.json"), "w"))
    json.dump(scan_lut(args.ref), open(os.path.join(gen, "scan_lut.json"), "w"))

    rng = np.random.default_rng(20261004)
    xs = np.concatenate([
        rng.integers(-2048, 2048, (96, 64)),
        # |x| <= 35000 is provably free of int32 overflow in the reference's own
        # arithmetic (max over multiplies of L1(operand)*K = 59914 < 2^31/35000),
        # so exact integers and the compiled reference's wrapping ints agree.
        rng.integers(-35000, 35001, (128, 64)),
        np.full((1, 64), 2047), np.full((1, 64), -2048), np.zeros((1, 64), dtype=np.int64),
        np.eye(64, dtype=np.int64) * 2047, np.eye(64, dtype=np.int64) * -2048,
    ]).astype(np.int64)
    ys = numeric_golden(body, xs)
    assert np.abs(ys).max() < 2 ** 31
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    np.savez_compressed(os.path.join(gold, "fdct64_vectors.npz"),
                        x=xs.astype(np.int32), y=ys.astype(np.int32))
    print("golden vectors:", xs.shape)

    # inverse transform (decoder side): od_bin_idct64, ffv2.c:4814-4948
    ibody = expanded_statements(args.ref, "od_bin_idct64")
    ilow = Lowerer()
    ilow.run(ibody)
    iir = compact(ilow)
    print("idct ops:", len(iir["ops"]), "regs:", iir["n_regs"])
    json.dump(iir, open(os.path.join(gen, "idct64_ir.json"), "w"))
    # golden: coefficient vectors = forward outputs of the golden inputs (plus some raw noise)
    cs = np.concatenate([ys[:224], rng.integers(-3000, 3001, (64, 64))]).astype(np.int64)
    rs = numeric_golden(ibody, cs)
    assert np.abs(rs).max() < 2 ** 31
    np.savez_compressed(os.path.join(gold, "idct64_vectors.npz"), y=cs.astype(np.int32), x=rs.astype(np.int32))
    print("inverse golden vectors:", cs.shape, "max |idct(fdct(x)) - x| on the first 224:",
          int(np.abs(rs[:224] - xs[:224]).max()))


if __name__ == "__main__":
    sys.exit(main())
