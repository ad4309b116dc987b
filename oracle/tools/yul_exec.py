"""A tiny interpreter for the straight-line Yul that snark-verifier generates,
used to run the REFERENCE's verifier where it lies
(/root/reference/proving-server/P256Verifier.yul) on a proof.

Oracle tooling (test infrastructure): it exists only in the build container —
nothing here is a copy of the reference; the Yul file is read at run time and
tests that use it skip when /root/reference is absent (e.g. on the GPU box).

EVM precompiles 0x5 (modexp), 0x6 (ecAdd), 0x7 (ecMul) are implemented with the
oracle's BN254 code; 0x8 (pairing) is replaced by the equivalent check with the
known trusted-setup secret: e(A, G2) * e(B, -[tau]G2) == 1  <=>  A == [tau] B
(SURVEY.md §0.3, Appendix B.4).
"""
import re
import sys

from zkoracle import curve as C
from zkoracle.field import P
from zkoracle.hashes import keccak256
from zkoracle.srs import TAU

M256 = (1 << 256) - 1
TOKEN = re.compile(r"\s*(:=|->|[{}(),]|0x[0-9a-fA-F]+|\d+|[A-Za-z_][A-Za-z_0-9]*(?::bool)?|\"[^\"]*\")")


def tokenize(src):
    pos, out = 0, []
    src = re.sub(r"//[^\n]*", "", src)
    while True:
        m = TOKEN.match(src, pos)
        if not m:
            if src[pos:].strip():
                raise SyntaxError("bad token at %r" % src[pos:pos + 40])
            return out
        t = m.group(1)
        out.append(t[:-5] if t.endswith(":bool") else t)
        pos = m.end()


class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else None

    def eat(self, x=None):
        v = self.t[self.i]
        if x is not None and v != x:
            raise SyntaxError("expected %s got %s" % (x, v))
        self.i += 1
        return v

    def block(self):
        self.eat("{")
        st = []
        while self.peek() != "}":
            st.append(self.stmt())
        self.eat("}")
        return ("block", st)

    def stmt(self):
        p = self.peek()
        if p == "{":
            return self.block()
        if p == "function":
            self.eat()
            name = self.eat()
            self.eat("(")
            args = []
            while self.peek() != ")":
                args.append(self.eat())
                if self.peek() == ",":
                    self.eat()
            self.eat(")")
            rets = []
            if self.peek() == "->":
                self.eat()
                rets.append(self.eat())
            return ("func", name, args, rets, self.block())
        if p == "let":
            self.eat()
            name = self.eat()
            if self.peek() == ":=":
                self.eat()
                return ("let", name, self.expr())
            return ("let", name, None)
        if p == "if":
            self.eat()
            c = self.expr()
            return ("if", c, self.block())
        if self.t[self.i + 1] == ":=":
            name = self.eat()
            self.eat(":=")
            return ("assign", name, self.expr())
        return ("expr", self.expr())

    def expr(self):
        t = self.eat()
        if t.startswith("0x"):
            return ("num", int(t, 16))
        if t[0].isdigit():
            return ("num", int(t))
        if self.peek() == "(":
            self.eat("(")
            args = []
            while self.peek() != ")":
                args.append(self.expr())
                if self.peek() == ",":
                    self.eat()
            self.eat(")")
            return ("call", t, args)
        return ("var", t)


class Halt(Exception):
    def __init__(self, reverted):
        self.reverted = reverted


class VM:
    def __init__(self, calldata, trace_mem=False):
        self.mem = bytearray(0x10000)
        self.cd = bytes(calldata) + bytes(64)
        self.scopes = [{}]
        self.funcs = {}
        self.precompile_calls = {5: 0, 6: 0, 7: 0, 8: 0}
        self.keccak_log = []

    # -- environment
    def get(self, n):
        for s in reversed(self.scopes):
            if n in s:
                return s[n]
        raise NameError(n)

    def set(self, n, v):
        for s in reversed(self.scopes):
            if n in s:
                s[n] = v
                return
        raise NameError(n)

    def mload(self, a):
        return int.from_bytes(self.mem[a:a + 32], "big")

    def mstore(self, a, v):
        self.mem[a:a + 32] = (v & M256).to_bytes(32, "big")

    # -- precompiles
    def staticcall(self, addr, ioff, ilen, ooff, olen):
        inp = bytes(self.mem[ioff:ioff + ilen])
        w = lambda i: int.from_bytes(inp[32 * i:32 * i + 32], "big")
        self.precompile_calls[addr] += 1
        try:
            return self._precompile(addr, inp, w, ilen, ooff, olen)
        except AssertionError:
            return 0  # the EVM precompile fails on malformed input

    def _precompile(self, addr, inp, w, ilen, ooff, olen):
        if addr == 5:
            assert (w(0), w(1), w(2)) == (32, 32, 32)
            out = pow(w(3), w(4), w(5)).to_bytes(32, "big")
        elif addr == 6:
            a, b = self.pt(w(0), w(1)), self.pt(w(2), w(3))
            out = self.enc(C.add(a, b))
        elif addr == 7:
            out = self.enc(C.mul(self.pt(w(0), w(1)), w(2)))
        elif addr == 8:
            assert ilen == 0x180
            A, B = self.pt(w(0), w(1)), self.pt(w(6), w(7))
            g2 = ((w(3), w(2)), (w(5), w(4)))
            ntg2 = ((w(9), w(8)), (w(11), w(10)))
            assert g2 == C.G2_GEN
            tg2 = C.g2_mul(C.G2_GEN, TAU)
            assert ntg2 == (tg2[0], ((-tg2[1][0]) % P, (-tg2[1][1]) % P)), "verifier's s_g2 is not -[tau]G2"
            ok = A == C.mul(B, TAU)
            out = int(ok).to_bytes(32, "big")
        else:
            raise ValueError(addr)
        self.mem[ooff:ooff + olen] = out[:olen]
        return 1

    @staticmethod
    def pt(x, y):
        if x == 0 and y == 0:
            return None
        assert C.is_on_curve((x, y))
        return (x, y)

    @staticmethod
    def enc(p):
        if p is None:
            return bytes(64)
        return p[0].to_bytes(32, "big") + p[1].to_bytes(32, "big")

    # -- evaluation
    def ev(self, e):
        k = e[0]
        if k == "num":
            return e[1]
        if k == "var":
            if e[1] in ("true", "false"):
                return int(e[1] == "true")
            return self.get(e[1])
        name, args = e[1], e[2]
        if name in self.funcs:
            return self.callf(name, [self.ev(a) for a in args])
        a = [self.ev(x) for x in args]
        if name == "mload":
            return self.mload(a[0])
        if name == "mstore":
            return self.mstore(a[0], a[1])
        if name == "mstore8":
            self.mem[a[0]] = a[1] & 0xFF
            return None
        if name == "calldataload":
            return int.from_bytes(self.cd[a[0]:a[0] + 32], "big")
        if name == "mulmod":
            return a[0] * a[1] % a[2]
        if name == "addmod":
            return (a[0] + a[1]) % a[2]
        if name == "sub":
            return (a[0] - a[1]) & M256
        if name == "add":
            return (a[0] + a[1]) & M256
        if name == "mod":
            return a[0] % a[1]
        if name == "eq":
            return int(a[0] == a[1])
        if name == "lt":
            return int(a[0] < a[1])
        if name == "and":
            return a[0] & a[1]
        if name == "not":
            return int(a[0] == 0)  # only used on the typed bool `success`
        if name == "gas":
            return M256
        if name == "keccak256":
            data = bytes(self.mem[a[0]:a[0] + a[1]])
            h = int.from_bytes(keccak256(data), "big")
            self.keccak_log.append((a[0], a[1], h))
            return h
        if name == "staticcall":
            return self.staticcall(a[1], a[2], a[3], a[4], a[5])
        if name == "revert":
            raise Halt(True)
        if name == "return":
            raise Halt(False)
        raise NameError(name)

    def callf(self, name, vals):
        _, _, params, rets, body = self.funcs[name]
        saved = self.scopes
        self.scopes = [dict(zip(params, vals))]
        for r in rets:
            self.scopes[0][r] = 0
        self.run(body)
        out = self.scopes[0][rets[0]] if rets else None
        self.scopes = saved
        return out

    def run(self, node):
        k = node[0]
        if k == "block":
            self.scopes.append({})
            for s in node[1]:
                self.run(s)
            self.scopes.pop()
        elif k == "func":
            self.funcs[node[1]] = node
        elif k == "let":
            self.scopes[-1][node[1]] = self.ev(node[2]) if node[2] is not None else 0
        elif k == "assign":
            self.set(node[1], self.ev(node[2]))
        elif k == "if":
            if self.ev(node[1]):
                self.run(node[2])
        elif k == "expr":
            self.ev(node[1])


def runtime_block(yul_text):
    """Returns the parsed `code { ... }` block of object "Runtime"."""
    i = yul_text.index('object "Runtime"')
    j = yul_text.index("code", i)
    toks = tokenize(yul_text[j + 4:])
    return Parser(toks).block()


def with_vk(yul_text, transcript_repr, commitments):
    """The same generated program with another circuit's verifying key: replaces, in memory, the
    transcript_repr literal and the 12 vk commitment literals (the 24 `mstore(addr, 0x<64 hex>)` after
    the generator (1, 2) and before the 8 G2 words) by the given ones.  Shape-dependent code is untouched,
    so the key must be of the shape the program was generated for (k=17, 4/1/1 columns)."""
    import re
    lit = list(re.finditer(r"mstore\(0x[0-9a-f]+, 0x([0-9a-f]{64})\)", yul_text))
    assert len(lit) == 2 + 24 + 8 and len(commitments) == 12
    flat = [c for pt in commitments for c in pt]
    out, last = [], 0
    for m, v in zip(lit[2:26], flat):
        out.append(yul_text[last:m.start(1)])
        out.append("%064x" % v)
        last = m.end(1)
    out.append(yul_text[last:])
    text = "".join(out)
    text, cnt = re.subn(r"mstore\(0x0, \d+\)", "mstore(0x0, %d)" % transcript_repr, text, count=1)
    assert cnt == 1
    return text


def run_verifier(yul_path, proof: bytes, vk=None):
    vm = VM(proof)
    text = open(yul_path).read()
    if vk is not None:
        text = with_vk(text, vk[0], vk[1])
    blk = runtime_block(text)
    try:
        vm.run(blk)
    except Halt as h:
        return (not h.reverted), vm
    return True, vm


if __name__ == "__main__":
    ok, vm = run_verifier(sys.argv[1], bytes.fromhex(open(sys.argv[2]).read().strip().removeprefix("0x")))
    print("accepted" if ok else "REJECTED", vm.precompile_calls)
