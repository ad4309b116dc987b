"""PLONKish proof system of the reference, restated: circuit shape, transcripts,
verifier (GWC + SHPLONK).  The prover restatement is in prover.py.

Oracle (test infrastructure) — see oracle/zkoracle/__init__.py.

Follows halo2_proofs (PSE fork; NOT under /root/reference) `plonk/verifier.rs`,
`plonk/{permutation,lookup,vanishing}/verifier.rs`,
`poly/kzg/multiopen/{gwc,shplonk}/verifier.rs`, `transcript.rs` and
snark-verifier `system/halo2/transcript/evm.rs`, as invoked by the reference at
halo2-circuits/src/ecc/ecdsa_p256.rs:439-445 (verify) and :461-467 (verify_evm).
PINNED for the Keccak/GWC path: `verify` accepts the reference's golden proof
(contracts/test/P256Account.t.sol:120-121) with the k=17 verifying key baked
into proving-server/P256Verifier.yul, and reproduces the Yul's challenges
(tests/test_oracle_verifier.py).  The pairing is replaced by the equivalent
check with the known trusted-setup secret tau.
The Blake2b/SHPLONK path is parity-unpinned (no reference bytes exist).
"""
import hashlib
from dataclasses import dataclass, field as dfield

from . import curve as C
from .field import DELTA, P, R, inv, omega
from .hashes import keccak256
from .srs import TAU

BLINDING_FACTORS = 6  # max(3, 4 queries of a gate column) + 2; l_last = row n-7 (P256Verifier.yul:308,773)

# forms of a gate's selector expression after halo2's compress_selectors (Shape.gate_sel)
SEL_PLAIN, SEL_FIRST_OF_TWO, SEL_SECOND_OF_TWO = 0, 1, 2


def selector_value(form, q):
    """The selector expression of a gate evaluated where its fixed column is q: q, q (2 - q) or q (1 - q)."""
    if form == SEL_PLAIN:
        return q % R
    return q * ((2 if form == SEL_FIRST_OF_TWO else 1) - q) % R


# ------------------------------------------------------------------ shape ---

@dataclass
class Shape:
    """Column/gate layout of halo2-lib's ECDSA circuit as a function of the JSON
    config (reference halo2-circuits/src/configs/bench_ecdsa.config)."""
    k: int
    num_advice: int            # A: gate columns
    num_lookup_advice: int     # L
    num_fixed: int             # F: constant columns
    lookup_bits: int = 0
    # gate columns (the last ones) whose selector is never enabled.  halo2's selector compression
    # (plonk/circuit/compress_selectors.rs `process` [RECALLED]) combines simple selectors that are never enabled on the
    # same row into ONE fixed column as long as the degree allows: with gate degree 3 and constraint-system degree 4 a
    # combination holds two selectors.  Used gate columns all start at row 0 and exclude each other; an all-false selector
    # excludes nobody, so the greedy pass puts the t-th never-enabled selector into the column of gate t: no fixed column
    # of its own (this is what makes the published proofs of the k <= 13 rows 1 / 2 / 3 evaluations shorter,
    # halo2-circuits/src/results/ecdsa_bench.csv:8-10) — and both gates of the pair change their
    # selector EXPRESSION: q (2 - q) for the used one (value 1 where q = 1), q (1 - q) for the never-enabled one (zero on
    # the domain, not as a polynomial: it contributes to h(X)).  `gate_sel[j]` = (fixed column, form).
    idle_gate_columns: int = 0

    def __post_init__(self):
        A, L, F = self.num_advice, self.num_lookup_advice, self.num_fixed
        U = self.idle_gate_columns
        assert 0 <= U < A and not (A == 1 and U)
        self.n = 1 << self.k
        self.single = A == 1
        self.n_gate = A
        self.n_lookup_cols = 0 if self.single else L
        self.n_adv = A + self.n_lookup_cols
        # fixed columns: constants, table, selectors
        self.fx_const = list(range(F))
        self.fx_table = F
        if self.single:
            # compress_selectors gives the selectors that occur in no gate (the complex q_lookup) their fixed columns
            # first, then the simple ones [RECALLED plonk/circuit/compress_selectors.rs `process`; zkoracle/vkrepr.py]
            self.fx_qlookup = F + 1
            self.fx_sel = [F + 2]          # q_enable
            self.n_fix = F + 3
        else:
            self.fx_sel = [F + 1 + j for j in range(A - U)] + [None] * U
            self.fx_qlookup = None
            self.n_fix = F + 1 + A - U
            assert 2 * U <= A, "more never-enabled selectors than used ones: they would pair up in columns of their own"
        # selector of gate j after compression: (fixed column, form); form 0: q, 1: q (2 - q), 2: q (1 - q)
        self.gate_sel = [(c, SEL_PLAIN) for c in self.fx_sel]
        for t in range(U):
            self.gate_sel[t] = (self.fx_sel[t], SEL_FIRST_OF_TWO)
            self.gate_sel[A - U + t] = (self.fx_sel[t], SEL_SECOND_OF_TWO)
        self.advice_queries = [(j, r) for j in range(A) for r in range(4)] + [(A + l, 0) for l in range(self.n_lookup_cols)]
        self.fixed_queries = [(f, 0) for f in range(self.n_fix)]
        self.perm_cols = [("fixed", f) for f in self.fx_const] + [("advice", j) for j in range(self.n_adv)]
        self.n_lookups = 1 if self.single else L
        self.degree = 5 if self.single else 4
        self.chunk_len = self.degree - 2
        self.n_chunks = (len(self.perm_cols) + self.chunk_len - 1) // self.chunk_len
        self.n_h = self.degree - 1
        self.ext_k = self.k + 2
        self.last_rot = -(BLINDING_FACTORS + 1)
        self.usable_rows = self.n - (BLINDING_FACTORS + 1)

    # number of G1 points / scalars in a proof
    def n_evals(self):
        return (len(self.advice_queries) + self.n_fix + 1 + len(self.perm_cols)
                + 3 * (self.n_chunks - 1) + 2 + 5 * self.n_lookups)

    def n_points_before_multiopen(self):
        return self.n_adv + 2 * self.n_lookups + self.n_chunks + self.n_lookups + 1 + self.n_h

    def gwc_sets(self):
        return 5 + (1 if self.n_chunks > 1 else 0)


@dataclass
class VerifyingKey:
    shape: Shape
    fixed_commitments: list
    permutation_commitments: list
    transcript_repr: int


# ------------------------------------------------------------ transcripts ---

class EvmTranscript:
    """snark-verifier EvmTranscript: Keccak-256 over a running buffer
    (rule pinned by reference proving-server/P256Verifier.yul:34,75-81,97-109)."""

    def __init__(self, proof=None):
        self.buf = b""
        self.proof = proof
        self.pos = 0
        self.out = bytearray()

    def common_scalar(self, s):
        self.buf += s.to_bytes(32, "big")

    def common_point(self, pt):
        self.buf += pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")

    def squeeze(self):
        data = self.buf if len(self.buf) != 32 else self.buf + b"\x01"
        h = keccak256(data)
        self.buf = h
        return int.from_bytes(h, "big") % R

    # reader
    def read_point(self):
        b = self.proof[self.pos:self.pos + 64]
        if len(b) != 64:
            raise ValueError("proof too short")
        self.pos += 64
        pt = (int.from_bytes(b[:32], "big"), int.from_bytes(b[32:], "big"))
        if not C.is_on_curve(pt) or pt == (0, 0):
            raise ValueError("invalid point")
        self.common_point(pt)
        return pt

    def read_scalar(self):
        b = self.proof[self.pos:self.pos + 32]
        if len(b) != 32:
            raise ValueError("proof too short")
        self.pos += 32
        s = int.from_bytes(b, "big")
        if s >= R:
            raise ValueError("non-canonical scalar")
        self.common_scalar(s)
        return s

    # writer
    def write_point(self, pt):
        if pt is None:
            raise ValueError("cannot write the identity to the transcript")
        self.common_point(pt)
        self.out += pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")

    def write_scalar(self, s):
        self.common_scalar(s)
        self.out += s.to_bytes(32, "big")

    def finalize(self):
        return bytes(self.out)

    def done(self):
        return self.pos == len(self.proof)


def _sqrt_fq(a):
    # P = 3 mod 4
    r = pow(a, (P + 1) // 4, P)
    return r if r * r % P == a % P else None


class Blake2bTranscript:
    """halo2_proofs Blake2bWrite/Blake2bRead with Challenge255 (transcript.rs):
    blake2b-512, personal "Halo2-Transcript"; prefix bytes 0 (challenge), 1 (point),
    2 (scalar); points hashed as x||y (32 B LE each) and written compressed."""

    def __init__(self, proof=None):
        self.st = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.proof = proof
        self.pos = 0
        self.out = bytearray()

    def common_scalar(self, s):
        self.st.update(b"\x02" + s.to_bytes(32, "little"))

    def common_point(self, pt):
        self.st.update(b"\x01" + pt[0].to_bytes(32, "little") + pt[1].to_bytes(32, "little"))

    def squeeze(self):
        self.st.update(b"\x00")
        return int.from_bytes(self.st.copy().digest(), "little") % R

    @staticmethod
    def compress(pt):
        b = bytearray(pt[0].to_bytes(32, "little"))
        b[31] |= (pt[1] & 1) << 7
        return bytes(b)

    def read_point(self):
        b = bytearray(self.proof[self.pos:self.pos + 32])
        if len(b) != 32:
            raise ValueError("proof too short")
        self.pos += 32
        sign = b[31] >> 7
        b[31] &= 0x7F
        x = int.from_bytes(b, "little")
        if x >= P:
            raise ValueError("invalid point")
        y = _sqrt_fq((x * x * x + 3) % P)
        if y is None:
            raise ValueError("invalid point")
        if (y & 1) != sign:
            y = P - y
        pt = (x, y)
        self.common_point(pt)
        return pt

    def read_scalar(self):
        b = self.proof[self.pos:self.pos + 32]
        if len(b) != 32:
            raise ValueError("proof too short")
        self.pos += 32
        s = int.from_bytes(b, "little")
        if s >= R:
            raise ValueError("non-canonical scalar")
        self.common_scalar(s)
        return s

    def write_point(self, pt):
        if pt is None:
            raise ValueError("cannot write the identity to the transcript")
        self.common_point(pt)
        self.out += self.compress(pt)

    def write_scalar(self, s):
        self.common_scalar(s)
        self.out += s.to_bytes(32, "little")

    def finalize(self):
        return bytes(self.out)

    def done(self):
        return self.pos == len(self.proof)


def make_transcript(kind, proof=None):
    return EvmTranscript(proof) if kind == "evm" else Blake2bTranscript(proof)


# --------------------------------------------------------------- helpers ----

def lagrange_evals_at(shape, x):
    """l_0(x), l_last(x), l_blind(x) (P256Verifier.yul:306-323,398-405)."""
    n, k = shape.n, shape.k
    w = omega(k)
    xn = pow(x, n, R)
    c = (xn - 1) * inv(n, R) % R

    def L(i):  # row index may be negative
        wi = pow(w, i, R)
        return wi * c % R * inv((x - wi) % R, R) % R

    l0 = L(0)
    l_last = L(-(BLINDING_FACTORS + 1))
    l_blind = sum(L(-i) for i in range(1, BLINDING_FACTORS + 1)) % R
    return l0, l_last, l_blind, xn


def msm_points(terms):
    """sum s_i * P_i for [(scalar, point)]"""
    acc = None
    for s, pt in terms:
        s %= R
        if s and pt is not None:
            acc = C.add(acc, C.mul(pt, s))
    return acc


def lagrange_interpolate(points, evals):
    """coefficients (low to high) of the polynomial through (points_i, evals_i)."""
    m = len(points)
    coeffs = [0] * m
    for j in range(m):
        # basis numerator prod_{i != j} (X - p_i)
        num = [1]
        den = 1
        for i in range(m):
            if i == j:
                continue
            num = [((num[t - 1] if t > 0 else 0) - points[i] * (num[t] if t < len(num) else 0)) % R for t in range(len(num) + 1)]
            den = den * (points[j] - points[i]) % R
        sc = evals[j] * inv(den, R) % R
        for t in range(m):
            coeffs[t] = (coeffs[t] + num[t] * sc) % R
    return coeffs


def eval_poly(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


def vanishing_eval(points, x):
    acc = 1
    for p in points:
        acc = acc * (x - p) % R
    return acc


# -------------------------------------------------------------- verifier ----

@dataclass
class Proof:
    advice: list = dfield(default_factory=list)
    lookups_permuted: list = dfield(default_factory=list)   # [(a', s')]
    perm_z: list = dfield(default_factory=list)
    lookup_z: list = dfield(default_factory=list)
    random: tuple = None
    h: list = dfield(default_factory=list)
    advice_evals: list = dfield(default_factory=list)
    fixed_evals: list = dfield(default_factory=list)
    random_eval: int = 0
    sigma_evals: list = dfield(default_factory=list)
    perm_evals: list = dfield(default_factory=list)          # per chunk (z, z_next, z_last|None)
    lookup_evals: list = dfield(default_factory=list)        # (z, z_next, a', a'_inv, s')
    challenges: dict = dfield(default_factory=dict)


def read_proof(vk, tr):
    sh = vk.shape
    pf = Proof()
    tr.common_scalar(vk.transcript_repr)
    pf.advice = [tr.read_point() for _ in range(sh.n_adv)]
    theta = tr.squeeze()
    pf.lookups_permuted = [(tr.read_point(), tr.read_point()) for _ in range(sh.n_lookups)]
    beta = tr.squeeze()
    gamma = tr.squeeze()
    pf.perm_z = [tr.read_point() for _ in range(sh.n_chunks)]
    pf.lookup_z = [tr.read_point() for _ in range(sh.n_lookups)]
    pf.random = tr.read_point()
    y = tr.squeeze()
    pf.h = [tr.read_point() for _ in range(sh.n_h)]
    x = tr.squeeze()
    pf.advice_evals = [tr.read_scalar() for _ in sh.advice_queries]
    pf.fixed_evals = [tr.read_scalar() for _ in sh.fixed_queries]
    pf.random_eval = tr.read_scalar()
    pf.sigma_evals = [tr.read_scalar() for _ in sh.perm_cols]
    for i in range(sh.n_chunks):
        z, zn = tr.read_scalar(), tr.read_scalar()
        zl = tr.read_scalar() if i != sh.n_chunks - 1 else None
        pf.perm_evals.append((z, zn, zl))
    for _ in range(sh.n_lookups):
        pf.lookup_evals.append(tuple(tr.read_scalar() for _ in range(5)))
    pf.challenges = dict(theta=theta, beta=beta, gamma=gamma, y=y, x=x)
    return pf


def expected_h_eval(vk, pf):
    """y-Horner over gate, permutation and lookup expressions, divided by x^n - 1
    (expressions pinned by P256Verifier.yul:406-552)."""
    sh = vk.shape
    ch = pf.challenges
    theta, beta, gamma, y, x = ch["theta"], ch["beta"], ch["gamma"], ch["y"], ch["x"]
    adv = {q: e for q, e in zip(sh.advice_queries, pf.advice_evals)}
    fix = {q[0]: e for q, e in zip(sh.fixed_queries, pf.fixed_evals)}
    l0, l_last, l_blind, xn = lagrange_evals_at(sh, x)
    active = (1 - l_last - l_blind) % R
    exprs = []
    # gates: q_j * (a + b*c - d)
    for j in range(sh.n_gate):
        a, b, c, d = (adv[(j, r)] for r in range(4))
        col, form = sh.gate_sel[j]
        exprs.append(selector_value(form, fix[col]) * ((a + b * c - d) % R) % R)
    # permutation argument
    col_eval = lambda col: fix[col[1]] if col[0] == "fixed" else adv[(col[1], 0)]
    pe = pf.perm_evals
    exprs.append(l0 * (1 - pe[0][0]) % R)
    zl = pe[-1][0]
    exprs.append(l_last * ((zl * zl - zl) % R) % R)
    for i in range(1, sh.n_chunks):
        exprs.append(l0 * ((pe[i][0] - pe[i - 1][2]) % R) % R)
    for i in range(sh.n_chunks):
        cols = sh.perm_cols[i * sh.chunk_len:(i + 1) * sh.chunk_len]
        sig = pf.sigma_evals[i * sh.chunk_len:(i + 1) * sh.chunk_len]
        left = pe[i][1]
        for col, s in zip(cols, sig):
            left = left * ((col_eval(col) + beta * s + gamma) % R) % R
        right = pe[i][0]
        cur = beta * x % R * pow(DELTA, i * sh.chunk_len, R) % R
        for col in cols:
            right = right * ((col_eval(col) + cur + gamma) % R) % R
            cur = cur * DELTA % R
        exprs.append(active * ((left - right) % R) % R)
    # lookups
    for l in range(sh.n_lookups):
        z, zn, ap, ap_inv, sp = pf.lookup_evals[l]
        if sh.single:
            inp = fix[sh.fx_qlookup] * adv[(0, 0)] % R
        else:
            inp = adv[(sh.n_gate + l, 0)]
        tab = fix[sh.fx_table]
        # theta-compression of single-expression vectors is the identity
        exprs.append(l0 * (1 - z) % R)
        exprs.append(l_last * ((z * z - z) % R) % R)
        left = zn * ((ap + beta) % R) % R * ((sp + gamma) % R) % R
        right = z * ((inp + beta) % R) % R * ((tab + gamma) % R) % R
        exprs.append(active * ((left - right) % R) % R)
        exprs.append(l0 * ((ap - sp) % R) % R)
        exprs.append(active * ((ap - sp) % R) % R * ((ap - ap_inv) % R) % R)
    acc = 0
    for e in exprs:
        acc = (acc * y + e) % R
    return acc * inv((xn - 1) % R, R) % R, xn


def build_queries(vk, pf):
    """(commitment, rotation, eval) in halo2's verifier order; a commitment is a G1
    point or ("h",) for the combined quotient commitment."""
    sh = vk.shape
    q = []
    for (col, rot), e in zip(sh.advice_queries, pf.advice_evals):
        q.append((("adv", col), rot, e))
    for i in range(sh.n_chunks):
        q.append((("z", i), 0, pf.perm_evals[i][0]))
        q.append((("z", i), 1, pf.perm_evals[i][1]))
    for i in reversed(range(sh.n_chunks - 1)):
        q.append((("z", i), sh.last_rot, pf.perm_evals[i][2]))
    for l in range(sh.n_lookups):
        z, zn, ap, ap_inv, sp = pf.lookup_evals[l]
        q.append((("lz", l), 0, z))
        q.append((("la", l), 0, ap))
        q.append((("ls", l), 0, sp))
        q.append((("la", l), -1, ap_inv))
        q.append((("lz", l), 1, zn))
    for (col, rot), e in zip(sh.fixed_queries, pf.fixed_evals):
        q.append((("fix", col), rot, e))
    for i, e in enumerate(pf.sigma_evals):
        q.append((("sigma", i), 0, e))
    return q


def commitment_point(vk, pf, key, xn):
    kind = key[0]
    if kind == "adv":
        return pf.advice[key[1]]
    if kind == "z":
        return pf.perm_z[key[1]]
    if kind == "lz":
        return pf.lookup_z[key[1]]
    if kind == "la":
        return pf.lookups_permuted[key[1]][0]
    if kind == "ls":
        return pf.lookups_permuted[key[1]][1]
    if kind == "fix":
        return vk.fixed_commitments[key[1]]
    if kind == "sigma":
        return vk.permutation_commitments[key[1]]
    if kind == "h":
        return msm_points([(pow(xn, i, R), h) for i, h in enumerate(pf.h)])
    if kind == "rand":
        return pf.random
    raise KeyError(key)


def verify(vk, proof: bytes, kind="evm", scheme=None, return_detail=False):
    """True iff the proof verifies.  kind: "evm" (Keccak, default GWC) or "blake2b"
    (default SHPLONK) — the pairings of the reference at ecdsa_p256.rs:439-445,461-467."""
    scheme = scheme or ("gwc" if kind == "evm" else "shplonk")
    sh = vk.shape
    tr = make_transcript(kind, proof)
    try:
        pf = read_proof(vk, tr)
        h_eval, xn = expected_h_eval(vk, pf)
        x = pf.challenges["x"]
        w = omega(sh.k)
        queries = build_queries(vk, pf)
        queries.append((("h",), 0, h_eval))
        queries.append((("rand",), 0, pf.random_eval))
        pt_of = lambda rot: x * pow(w, rot, R) % R
        cache = {}

        def cpt(key):
            if key not in cache:
                cache[key] = commitment_point(vk, pf, key, xn)
            return cache[key]

        if scheme == "gwc":
            v = tr.squeeze()
            sets = []  # [(rot, [(key, eval)])] by first appearance of the point
            for key, rot, e in queries:
                for s in sets:
                    if s[0] == rot:
                        s[1].append((key, e))
                        break
                else:
                    sets.append((rot, [(key, e)]))
            ws = [tr.read_point() for _ in sets]
            u = tr.squeeze()
            pf.challenges.update(v=v, u=u)
            left, right = [], []
            eval_multi = 0
            pu = 1
            for (rot, qs), wi in zip(sets, ws):
                z = pt_of(rot)
                pv = 1
                eb = 0
                for key, e in qs:
                    right.append((pu * pv, cpt(key)))
                    eb = (eb + pv * e) % R
                    pv = pv * v % R
                eval_multi = (eval_multi + pu * eb) % R
                right.append((pu * z, wi))
                left.append((pu, wi))
                pu = pu * u % R
            right.append((-eval_multi, C.G1_GEN))
            lhs, rhs = msm_points(left), msm_points(right)
            ok = rhs == (C.mul(lhs, TAU) if lhs is not None else None)  # e(lhs, [tau]G2) == e(rhs, G2)
        else:
            # rotation sets keyed by the set of rotations; points ordered as field elements (BTreeSet<Fr>)
            com_rots = []  # [(key, set(rots))] by first appearance of the commitment
            evals = {}
            for key, rot, e in queries:
                evals[(key, rot)] = e
                for cr in com_rots:
                    if cr[0] == key:
                        cr[1].add(rot)
                        break
                else:
                    com_rots.append((key, {rot}))
            rsets = []  # [(frozenset rots, [keys])]
            for key, rots in com_rots:
                fr = frozenset(rots)
                for rs in rsets:
                    if rs[0] == fr:
                        rs[1].append(key)
                        break
                else:
                    rsets.append((fr, [key]))
            all_rots = sorted({r for _, r, _ in queries}, key=pt_of)
            yc = tr.squeeze()
            v = tr.squeeze()
            h1 = tr.read_point()
            u = tr.squeeze()
            h2 = tr.read_point()
            pf.challenges.update(shplonk_y=yc, v=v, u=u)
            terms = []
            r_outer = 0
            z0 = z0_diff_inv = 0
            pv = 1
            for i, (rots, keys) in enumerate(rsets):
                rl = sorted(rots, key=pt_of)
                pts = [pt_of(r) for r in rl]
                diffs = [pt_of(r) for r in all_rots if r not in rots]
                zd = vanishing_eval(diffs, u)
                if i == 0:
                    z0 = vanishing_eval(pts, u)
                    z0_diff_inv = inv(zd, R)
                    zd = 1
                else:
                    zd = zd * z0_diff_inv % R
                py = 1
                r_inner = 0
                for key in keys:
                    rx = lagrange_interpolate(pts, [evals[(key, r)] for r in rl])
                    r_inner = (r_inner + py * eval_poly(rx, u)) % R
                    terms.append((py * pv % R * zd, cpt(key)))
                    py = py * yc % R
                r_outer = (r_outer + pv * r_inner % R * zd) % R
                pv = pv * v % R
            terms.append((-r_outer, C.G1_GEN))
            terms.append((-z0, h1))
            terms.append((u, h2))
            ok = msm_points(terms) == C.mul(h2, TAU)
        ok = ok and tr.done()
    except ValueError:
        if return_detail:
            raise
        return False
    return (ok, pf) if return_detail else ok
