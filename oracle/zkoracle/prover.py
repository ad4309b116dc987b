"""Keygen and create_proof of the reference, restated in plain Python for small k.

Oracle (test infrastructure) — see oracle/zkoracle/__init__.py.

Follows halo2_proofs (PSE fork; NOT under /root/reference) `plonk/keygen.rs`,
`plonk/prover.rs`, `plonk/permutation/{keygen,prover}.rs`, `plonk/lookup/prover.rs`,
`plonk/vanishing/prover.rs`, `plonk/evaluation.rs`,
`poly/kzg/multiopen/{gwc,shplonk}/prover.rs`, reached from the reference at
halo2-circuits/src/ecc/ecdsa_p256.rs:259-260 (keygen_vk/keygen_pk) and
:366-373, :416-423, :555-562 (create_proof).  Structure and RNG draw order as
SURVEY.md §3.3 / App. A.3; every commitment is computed with the known
trusted-setup secret (commit(f) = [f(tau)]G1), i.e. without any
MSM — which makes this an independent check of the engine's MSM/NTT path.

Anchoring: proofs made here are accepted by zkoracle.plonk.verify, the verifier
pinned by the reference's golden proof.  The Blake2b/SHPLONK byte layout is
parity-unpinned (no reference bytes exist).

Everything is canonical Python ints; vectors are lists of length n.
"""
import hashlib

from . import curve as C
from .field import DELTA, R, ZETA, batch_inv, inv, omega
from .plonk import BLINDING_FACTORS, Shape, VerifyingKey, make_transcript, selector_value
from .srs import TAU, lagrange_at


# ------------------------------------------------------------- poly utils ---

def ntt(a, w):
    """out[i] = sum_j a[j] w^(ij); len(a) a power of two (recursive radix-2)."""
    n = len(a)
    if n == 1:
        return a[:]
    w2 = w * w % R
    ev, od = ntt(a[0::2], w2), ntt(a[1::2], w2)
    out = [0] * n
    t = 1
    h = n // 2
    for i in range(h):
        x = t * od[i] % R
        out[i] = (ev[i] + x) % R
        out[i + h] = (ev[i] - x) % R
        t = t * w % R
    return out


def lagrange_to_coeff(v, k):
    n = 1 << k
    ninv = inv(n, R)
    return [x * ninv % R for x in ntt(v, inv(omega(k), R))]


def coeff_to_extended(c, k, ext_k):
    """evaluations of c(X) at zeta * w_ext^i (EvaluationDomain::coeff_to_extended)."""
    N = 1 << ext_k
    zp = [1, ZETA, ZETA * ZETA % R]
    a = [c[i] * zp[i % 3] % R if i < len(c) else 0 for i in range(N)]
    return ntt(a, omega(ext_k))


def extended_to_coeff(e, ext_k):
    N = 1 << ext_k
    ninv = inv(N, R)
    a = ntt(e, inv(omega(ext_k), R))
    zi = [1, ZETA * ZETA % R, ZETA]
    return [a[i] * ninv % R * zi[i % 3] % R for i in range(N)]


def eval_poly(c, x):
    acc = 0
    for ci in reversed(c):
        acc = (acc * x + ci) % R
    return acc


def kate_division(c, z):
    """(c(X) - c(z)) / (X - z) for coefficient list c; returns len(c)-1 coefficients."""
    q = [0] * (len(c) - 1)
    carry = 0
    for i in range(len(c) - 1, 0, -1):
        carry = (c[i] + carry * z) % R
        q[i - 1] = carry
    return q


def commit_coeff(c):
    return C.mul(C.G1_GEN, eval_poly(c, TAU))


class Committer:
    def __init__(self, k):
        self.lag = lagrange_at(k, TAU)

    def lagrange(self, v):
        return C.mul(C.G1_GEN, sum(a * b for a, b in zip(v, self.lag)) % R)


# ----------------------------------------------------------------- keygen ---

class Circuit:
    """What synthesize() leaves behind: fixed columns, copy constraints, advice."""

    def __init__(self, shape: Shape, fixed, copies, advice):
        self.shape = shape
        self.fixed = fixed      # [n_fix][n] ints
        self.copies = copies    # [((colidx_in_perm_cols, row), (colidx, row))]
        self.advice = advice    # [n_adv][n] ints (rows >= usable ignored: blinded)


def build_sigma(shape: Shape, copies):
    """permutation::keygen::Assembly: cycles merged per copy; sigma_c[row] = delta^c' * w^row'."""
    n, m = shape.n, len(shape.perm_cols)
    mapping = [[(c, r) for r in range(n)] for c in range(m)]
    aux = [[(c, r) for r in range(n)] for c in range(m)]
    sizes = [[1] * n for _ in range(m)]
    for (lc, lr), (rc, rr) in copies:
        assert lr < shape.usable_rows and rr < shape.usable_rows
        lcy, rcy = aux[lc][lr], aux[rc][rr]
        if lcy == rcy:
            continue
        if sizes[lcy[0]][lcy[1]] < sizes[rcy[0]][rcy[1]]:
            lcy, rcy = rcy, lcy
            (lc, lr), (rc, rr) = (rc, rr), (lc, lr)
        sizes[lcy[0]][lcy[1]] += sizes[rcy[0]][rcy[1]]
        i = rcy
        while True:
            aux[i[0]][i[1]] = lcy
            i = mapping[i[0]][i[1]]
            if i == rcy:
                break
        mapping[lc][lr], mapping[rc][rr] = mapping[rc][rr], mapping[lc][lr]
    w = omega(shape.k)
    wp = [1] * n
    for i in range(1, n):
        wp[i] = wp[i - 1] * w % R
    dp = [pow(DELTA, c, R) for c in range(m)]
    return [[dp[mapping[c][r][0]] * wp[mapping[c][r][1]] % R for r in range(n)] for c in range(m)]


def transcript_repr(shape: Shape, fixed_commitments, permutation_commitments):
    """halo2's vk.transcript_repr: the Blake2b hash of the pinned verifying key's Debug rendering (zkoracle/vkrepr.py,
    pinned by the reference's k = 17 value, P256Verifier.yul:34), for every shape — never-enabled gate columns included
    (their combined selectors are rendered as compress_selectors builds them)."""
    from . import vkrepr
    return vkrepr.transcript_repr(shape, fixed_commitments, permutation_commitments)


class ProvingKey:
    def __init__(self, shape, fixed, sigma, vk):
        self.shape, self.fixed, self.sigma, self.vk = shape, fixed, sigma, vk


def keygen(circuit: Circuit):
    sh = circuit.shape
    cm = Committer(sh.k)
    sigma = build_sigma(sh, circuit.copies)
    fc = [cm.lagrange(col) for col in circuit.fixed]
    pc = [cm.lagrange(col) for col in sigma]
    vk = VerifyingKey(sh, fc, pc, transcript_repr(sh, fc, pc))
    return ProvingKey(sh, circuit.fixed, sigma, vk)


# ----------------------------------------------------------------- prover ---

def permute_expression_pair(inp, tab, usable, rng):
    """lookup::prover::permute_expression_pair."""
    a = sorted(inp[:usable])
    left = {}
    for t in tab[:usable]:
        left[t] = left.get(t, 0) + 1
    s = [0] * usable
    repeated = []
    for row, v in enumerate(a):
        if row == 0 or v != a[row - 1]:
            s[row] = v
            if left.get(v, 0) <= 0:
                raise ValueError("lookup input not in table (ConstraintSystemFailure)")
            left[v] -= 1
        else:
            repeated.append(row)
    for t in sorted(left):
        for _ in range(left[t]):
            s[repeated.pop()] = t
    assert not repeated
    a = a + [rng.fr() for _ in range(BLINDING_FACTORS + 1)]
    s = s + [rng.fr() for _ in range(BLINDING_FACTORS + 1)]
    return a, s


def create_proof(pk: ProvingKey, advice, rng, kind="evm", scheme=None, trace=None):
    """Returns proof bytes.  `rng` provides .fr() (one Fr::random per call).
    `trace` (dict) receives intermediate values for operator-level parity tests."""
    scheme = scheme or ("gwc" if kind == "evm" else "shplonk")
    sh = pk.shape
    n, k, bf = sh.n, sh.k, BLINDING_FACTORS
    w = omega(k)
    cm = Committer(k)
    tr = make_transcript(kind)
    tr.common_scalar(pk.vk.transcript_repr)
    trace = {} if trace is None else trace

    # -- 1. advice: blind the last bf+1 rows, commit
    adv = [list(col) for col in advice]
    for col in adv:
        for r in range(sh.usable_rows, n):
            col[r] = rng.fr()
    for _ in adv:
        rng.fr()  # advice blinds (unused by KZG, still drawn)
    for col in adv:
        tr.write_point(cm.lagrange(col))
    theta = tr.squeeze()

    # -- 2. lookups: permuted input / table
    fixed = pk.fixed
    lk = []
    for l in range(sh.n_lookups):
        if sh.single:
            inp = [fixed[sh.fx_qlookup][i] * adv[0][i] % R for i in range(n)]
        else:
            inp = adv[sh.n_gate + l][:]
        tab = fixed[sh.fx_table][:]
        ap, sp = permute_expression_pair(inp, tab, sh.usable_rows, rng)
        rng.fr()
        rng.fr()  # two commitment blinds
        tr.write_point(cm.lagrange(ap))
        tr.write_point(cm.lagrange(sp))
        lk.append(dict(inp=inp, tab=tab, ap=ap, sp=sp))
    beta = tr.squeeze()
    gamma = tr.squeeze()

    # -- 3. permutation grand products
    def col_values(col):
        return fixed[col[1]] if col[0] == "fixed" else adv[col[1]]

    zs = []
    last_z = 1
    deltaomega0 = 1
    wp = [1] * n
    for i in range(1, n):
        wp[i] = wp[i - 1] * w % R
    for ci in range(sh.n_chunks):
        cols = sh.perm_cols[ci * sh.chunk_len:(ci + 1) * sh.chunk_len]
        sig = pk.sigma[ci * sh.chunk_len:(ci + 1) * sh.chunk_len]
        den = [1] * n
        for col, s in zip(cols, sig):
            v = col_values(col)
            den = [d * ((beta * s[i] + gamma + v[i]) % R) % R for i, d in enumerate(den)]
        frac = batch_inv(den, R)
        for col in cols:
            v = col_values(col)
            frac = [f * ((deltaomega0 * wp[i] % R * beta + gamma + v[i]) % R) % R for i, f in enumerate(frac)]
            deltaomega0 = deltaomega0 * DELTA % R
        z = [last_z]
        for row in range(1, n):
            z.append(z[row - 1] * frac[row - 1] % R)
        for r in range(n - bf, n):
            z[r] = rng.fr()
        last_z = z[n - (bf + 1)]
        rng.fr()  # blind
        tr.write_point(cm.lagrange(z))
        zs.append(z)

    # -- 4. lookup grand products
    for d in lk:
        den = [(beta + d["ap"][i]) % R * ((gamma + d["sp"][i]) % R) % R for i in range(n)]
        frac = batch_inv(den, R)
        frac = [frac[i] * ((d["inp"][i] + beta) % R) % R * ((d["tab"][i] + gamma) % R) % R for i in range(n)]
        z = [1]
        for i in range(n - bf - 1):
            z.append(z[-1] * frac[i] % R)
        z = z[:n - bf] + [rng.fr() for _ in range(bf)]
        rng.fr()  # blind
        tr.write_point(cm.lagrange(z))
        d["z"] = z

    # -- 5. vanishing: random polynomial
    random_poly = [rng.fr() for _ in range(n)]
    rng.fr()  # blind
    tr.write_point(commit_coeff(random_poly))
    y = tr.squeeze()

    # -- 6. quotient h(X) on the extended coset
    ext_k, N = sh.ext_k, 1 << sh.ext_k
    step = 1 << (ext_k - k)
    coeff = lambda v: lagrange_to_coeff(v, k)
    ext = lambda c: coeff_to_extended(c, k, ext_k)
    adv_c = [coeff(c) for c in adv]
    fix_c = [coeff(c) for c in fixed]
    sig_c = [coeff(c) for c in pk.sigma]
    z_c = [coeff(z) for z in zs]
    for d in lk:
        d["ap_c"], d["sp_c"], d["z_c"] = coeff(d["ap"]), coeff(d["sp"]), coeff(d["z"])
    adv_e = [ext(c) for c in adv_c]
    fix_e = [ext(c) for c in fix_c]
    sig_e = [ext(c) for c in sig_c]
    z_e = [ext(c) for c in z_c]
    unit = lambda rows: [1 if i in rows else 0 for i in range(n)]
    l0_e = ext(coeff(unit({0})))
    llast_e = ext(coeff(unit({n - bf - 1})))
    lblind_e = ext(coeff(unit(set(range(n - bf, n)))))
    wext = omega(ext_k)
    xs = [ZETA] * N  # coset points zeta * wext^i
    for i in range(1, N):
        xs[i] = xs[i - 1] * wext % R
    rot = lambda vec, i, r: vec[(i + r * step) % N]
    col_e = lambda col: fix_e[col[1]] if col[0] == "fixed" else adv_e[col[1]]
    hvals = [0] * N
    for i in range(N):
        acc = 0

        def push(e):
            nonlocal acc
            acc = (acc * y + e) % R

        l0, ll, lb = l0_e[i], llast_e[i], lblind_e[i]
        active = (1 - ll - lb) % R
        for j in range(sh.n_gate):
            a, b, c, d4 = (rot(adv_e[j], i, r) for r in range(4))
            col, form = sh.gate_sel[j]
            push(selector_value(form, fix_e[col][i]) * ((a + b * c - d4) % R) % R)
        push(l0 * (1 - z_e[0][i]) % R)
        zl = z_e[-1][i]
        push(ll * ((zl * zl - zl) % R) % R)
        for ci in range(1, sh.n_chunks):
            push(l0 * ((z_e[ci][i] - rot(z_e[ci - 1], i, sh.last_rot)) % R) % R)
        for ci in range(sh.n_chunks):
            cols = sh.perm_cols[ci * sh.chunk_len:(ci + 1) * sh.chunk_len]
            left = rot(z_e[ci], i, 1)
            for off, col in enumerate(cols):
                left = left * ((col_e(col)[i] + beta * sig_e[ci * sh.chunk_len + off][i] + gamma) % R) % R
            right = z_e[ci][i]
            cur = beta * xs[i] % R * pow(DELTA, ci * sh.chunk_len, R) % R
            for col in cols:
                right = right * ((col_e(col)[i] + cur + gamma) % R) % R
                cur = cur * DELTA % R
            push(active * ((left - right) % R) % R)
        for l, d in enumerate(lk):
            if "z_e" not in d:
                d["ap_e"], d["sp_e"], d["z_e"] = ext(d["ap_c"]), ext(d["sp_c"]), ext(d["z_c"])
            zc, zn = d["z_e"][i], rot(d["z_e"], i, 1)
            ap, apm, sp = d["ap_e"][i], rot(d["ap_e"], i, -1), d["sp_e"][i]
            if sh.single:
                inp = fix_e[sh.fx_qlookup][i] * adv_e[0][i] % R
            else:
                inp = adv_e[sh.n_gate + l][i]
            tab = fix_e[sh.fx_table][i]
            push(l0 * (1 - zc) % R)
            push(ll * ((zc * zc - zc) % R) % R)
            push(active * ((zn * ((ap + beta) % R) % R * ((sp + gamma) % R) - zc * ((inp + beta) % R) % R * ((tab + gamma) % R)) % R) % R)
            push(l0 * ((ap - sp) % R) % R)
            push(active * ((ap - sp) % R) % R * ((ap - apm) % R) % R)
        hvals[i] = acc
    # divide by X^n - 1 on the coset: (zeta * wext^i)^n - 1 has period `step`
    tinv = [inv((pow(xs[i], n, R) - 1) % R, R) for i in range(step)]
    hvals = [hv * tinv[i % step] % R for i, hv in enumerate(hvals)]
    h_coeff = extended_to_coeff(hvals, ext_k)
    assert all(c == 0 for c in h_coeff[n * sh.n_h:]), "quotient degree too high: constraints not satisfied"
    h_pieces = [h_coeff[i * n:(i + 1) * n] for i in range(sh.n_h)]
    for _ in h_pieces:
        rng.fr()  # h blinds
    for hp in h_pieces:
        tr.write_point(commit_coeff(hp))
    x = tr.squeeze()
    trace.update(theta=theta, beta=beta, gamma=gamma, y=y, x=x, h_coeff=h_coeff, adv=adv, zs=zs, lk=lk,
                 random_poly=random_poly)

    # -- 7. evaluations
    xr = lambda r: x * pow(w, r, R) % R
    evals = {}

    def ev(tag, poly, r):
        e = eval_poly(poly, xr(r))
        evals[(tag, r)] = e
        return e

    for col, r in sh.advice_queries:
        tr.write_scalar(ev(("adv", col), adv_c[col], r))
    for col, r in sh.fixed_queries:
        tr.write_scalar(ev(("fix", col), fix_c[col], r))
    xn = pow(x, n, R)
    h_comb = [0] * n
    for hp in reversed(h_pieces):
        h_comb = [(hc * xn + p) % R for hc, p in zip(h_comb, hp)]
    tr.write_scalar(ev(("rand",), random_poly, 0))
    for i, s in enumerate(sig_c):
        tr.write_scalar(ev(("sigma", i), s, 0))
    for ci in range(sh.n_chunks):
        tr.write_scalar(ev(("z", ci), z_c[ci], 0))
        tr.write_scalar(ev(("z", ci), z_c[ci], 1))
        if ci != sh.n_chunks - 1:
            tr.write_scalar(ev(("z", ci), z_c[ci], sh.last_rot))
    for l, d in enumerate(lk):
        tr.write_scalar(ev(("lz", l), d["z_c"], 0))
        tr.write_scalar(ev(("lz", l), d["z_c"], 1))
        tr.write_scalar(ev(("la", l), d["ap_c"], 0))
        tr.write_scalar(ev(("la", l), d["ap_c"], -1))
        tr.write_scalar(ev(("ls", l), d["sp_c"], 0))
    ev(("h",), h_comb, 0)

    # -- 8. multi-open: queries in prover order (same as the verifier's)
    polys = {("h",): h_comb, ("rand",): random_poly}
    for j, c in enumerate(adv_c):
        polys[("adv", j)] = c
    for j, c in enumerate(fix_c):
        polys[("fix", j)] = c
    for j, c in enumerate(sig_c):
        polys[("sigma", j)] = c
    for j, c in enumerate(z_c):
        polys[("z", j)] = c
    for l, d in enumerate(lk):
        polys[("lz", l)], polys[("la", l)], polys[("ls", l)] = d["z_c"], d["ap_c"], d["sp_c"]
    queries = [(("adv", col), r) for col, r in sh.advice_queries]
    for ci in range(sh.n_chunks):
        queries += [(("z", ci), 0), (("z", ci), 1)]
    for ci in reversed(range(sh.n_chunks - 1)):
        queries.append((("z", ci), sh.last_rot))
    for l in range(sh.n_lookups):
        queries += [(("lz", l), 0), (("la", l), 0), (("ls", l), 0), (("la", l), -1), (("lz", l), 1)]
    queries += [(("fix", col), r) for col, r in sh.fixed_queries]
    queries += [(("sigma", i), 0) for i in range(len(sig_c))]
    queries += [(("h",), 0), (("rand",), 0)]

    if scheme == "gwc":
        v = tr.squeeze()
        sets = []
        for key, r in queries:
            for s in sets:
                if s[0] == r:
                    s[1].append(key)
                    break
            else:
                sets.append((r, [key]))
        for r, keys in sets:
            pb = [0] * n
            eb = 0
            pv = 1
            for key in keys:
                p = polys[key]
                pb = [(a + pv * b) % R for a, b in zip(pb, p)]
                eb = (eb + pv * evals[(key, r)]) % R
                pv = pv * v % R
            pb[0] = (pb[0] - eb) % R
            tr.write_point(commit_coeff(kate_division(pb, xr(r))))
    else:
        com_rots = []
        for key, r in queries:
            for cr in com_rots:
                if cr[0] == key:
                    cr[1].add(r)
                    break
            else:
                com_rots.append((key, {r}))
        rsets = []
        for key, rots in com_rots:
            fr = frozenset(rots)
            for rs in rsets:
                if rs[0] == fr:
                    rs[1].append(key)
                    break
            else:
                rsets.append((fr, [key]))
        all_rots = sorted({r for _, r in queries}, key=xr)
        yc = tr.squeeze()
        v = tr.squeeze()
        from .plonk import lagrange_interpolate, vanishing_eval

        def div_by_vanishing(p, pts):
            for z in pts:
                p = kate_division(p, z)
            return p + [0] * (n - len(p))

        low = {}
        hx = [0] * n
        pv = 1
        for rots, keys in rsets:
            rl = sorted(rots, key=xr)
            pts = [xr(r) for r in rl]
            nx = [0] * n
            py = 1
            for key in keys:
                rxp = lagrange_interpolate(pts, [evals[(key, r)] for r in rl])
                low[key] = rxp
                num = polys[key][:]
                for t, c in enumerate(rxp):
                    num[t] = (num[t] - c) % R
                nx = [(a + py * b) % R for a, b in zip(nx, num)]
                py = py * yc % R
            q = div_by_vanishing(nx, pts)
            hx = [(a + pv * b) % R for a, b in zip(hx, q)]
            pv = pv * v % R
        tr.write_point(commit_coeff(hx))
        u = tr.squeeze()
        lx = [0] * n
        pv = 1
        z_diffs = []
        for rots, keys in rsets:
            diffs = [xr(r) for r in all_rots if r not in rots]
            zi = vanishing_eval(diffs, u)
            z_diffs.append(zi)
            inner = [0] * n
            py = 1
            for key in keys:
                p = polys[key][:]
                p[0] = (p[0] - eval_poly(low[key], u)) % R
                inner = [(a + py * b) % R for a, b in zip(inner, p)]
                py = py * yc % R
            lx = [(a + pv * zi % R * b) % R for a, b in zip(lx, inner)]
            pv = pv * v % R
        zt = vanishing_eval([xr(r) for r in all_rots], u)
        lx = [(a - zt * b) % R for a, b in zip(lx, hx)]
        assert eval_poly(lx, u) == 0
        hx2 = kate_division(lx, u)
        z0inv = inv(z_diffs[0], R)
        hx2 = [c * z0inv % R for c in hx2]
        tr.write_point(commit_coeff(hx2))
    return tr.finalize()
