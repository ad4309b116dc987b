"""GPU proof-level parity: the device prover (zk_keygen / zk_prove through the C-ABI)
produces, from the same witness and the same ChaCha20 stream, byte-identical proofs to
the oracle's restated create_proof, for every BASELINE column shape (scaled to small
k), both transcripts; at k=17 / k=19 the proofs are checked by the oracle verifier that
the reference's golden proof pins."""
import os

import numpy as np
import pytest

import webauthn_halo2_amd as zk
from webauthn_halo2_amd import engine as E
from zkoracle import cops, plonk, prover
from zkoracle.hashes import ChaCha20Rng

pytestmark = pytest.mark.gpu

SHAPES = {
    "k19like": (1, 1, 1, 7, 6),
    "k17like": (4, 1, 1, 7, 5),
    "k18like": (2, 1, 1, 6, 4),
    "wide": (3, 2, 2, 8, 6),
    "idle": (5, 2, 2, 7, 5, 2),  # two trailing gate columns never enabled (the k <= 13 bench rows)
    # k >= 10: the SRS window tables exist, so commitments go through the column-batched MSM passes (and the
    # batched transforms / grand products / divisions) that the full-size proofs use
    "k10batched": (3, 2, 1, 10, 8),
    "k10single": (1, 1, 1, 10, 9),
    # more than 32 terms in every group of the quotient's y-combination (gate terms, l_0 terms, active-row terms): the
    # lazy sums of quotient.hip are folded back below 2p (the k <= 13 bench rows do that at full size)
    "manycols": (36, 12, 2, 7, 5),
    # more than 40 polynomials in one rotation set over >= 256 rows: the multi-open's linear combinations go through the
    # argument-list kernel (lincomb_terms), the quotient through the lanes-per-row kernel
    "manycols_k8": (44, 6, 2, 8, 6),
}
KIND = {"evm": E.ZK_TRANSCRIPT_EVM, "blake2b": E.ZK_TRANSCRIPT_BLAKE2B}


def setup(engine, A, L, F, k, lb, seed=0x5EED0019, worst=False, idle=0):
    p = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb,
                                 idle_gate_columns=idle)
    asg = zk.circuit.synthesize(p, seed, worst_case=worst)
    engine.srs_setup(k)
    fixed = np.stack([asg.to_limbs(c) for c in asg.fixed])
    pk = engine.keygen(p, fixed, asg.copies)
    polys = []
    for col in asg.advice:
        h = engine.poly(1 << k)
        engine.upload_canonical(h, asg.to_limbs(col))
        polys.append(h)
    return p, asg, pk, polys


def product_vk(engine, pk, shape):
    fc, pc, tr = engine.vk_export(pk)
    return plonk.VerifyingKey(shape, cops.affine_arr_to_ints(fc), cops.affine_arr_to_ints(pc), cops.fr_ints(tr.reshape(1, 4))[0])


@pytest.mark.parametrize("name", list(SHAPES))
def test_proofs_byte_identical_to_oracle(engine, name):
    A, L, F, k, lb, idle = (SHAPES[name] + (0,))[:6]
    p, asg, pk, polys = setup(engine, A, L, F, k, lb, idle=idle)
    sh = plonk.Shape(k, A, L, F, lb, idle)
    opk = prover.keygen(prover.Circuit(sh, asg.fixed, asg.copies, asg.advice))
    vk = product_vk(engine, pk, sh)
    # keygen parity: commitments (MSM on device vs tau-oracle) and transcript_repr
    assert vk.fixed_commitments == opk.vk.fixed_commitments
    assert vk.permutation_commitments == opk.vk.permutation_commitments
    assert vk.transcript_repr == opk.vk.transcript_repr
    seed = bytes(range(32))
    for kind in ("evm", "blake2b"):
        for scheme in (E.ZK_SCHEME_DEFAULT,):
            got = engine.prove(pk, polys, seed, KIND[kind], scheme)
            want = prover.create_proof(opk, asg.advice, ChaCha20Rng(seed), kind)
            assert got == want, (name, kind)
            assert len(got) == engine.proof_size(pk, KIND[kind], scheme)
            assert plonk.verify(vk, got, kind)
    # the two non-reference pairings too
    got = engine.prove(pk, polys, seed, E.ZK_TRANSCRIPT_EVM, E.ZK_SCHEME_SHPLONK)
    assert got == prover.create_proof(opk, asg.advice, ChaCha20Rng(seed), "evm", "shplonk")
    got = engine.prove(pk, polys, seed, E.ZK_TRANSCRIPT_BLAKE2B, E.ZK_SCHEME_GWC)
    assert got == prover.create_proof(opk, asg.advice, ChaCha20Rng(seed), "blake2b", "gwc")
    for h in polys:
        h.free()
    engine.pk_free(pk)


def test_batch_inversion_fallback_same_bytes(engine):
    """The grand products normally use one inversion per product (prefix x suffix scans); a zero
    denominator sends them down the batch-inversion path.  Both must give the same proof."""
    A, L, F, k, lb = SHAPES["k17like"]
    _, asg, pk, polys = setup(engine, A, L, F, k, lb)
    seed = b"\x21" * 32
    a = engine.prove(pk, polys, seed, E.ZK_TRANSCRIPT_EVM)
    engine.set_option(E.ZK_OPT_GP_BATCH_INVERT, 1)
    try:
        b = engine.prove(pk, polys, seed, E.ZK_TRANSCRIPT_EVM)
    finally:
        engine.set_option(E.ZK_OPT_GP_BATCH_INVERT, 0)
    assert a == b
    for h in polys:
        h.free()
    engine.pk_free(pk)


def test_worst_case_witness_and_second_seed(engine):
    A, L, F, k, lb = SHAPES["k17like"]
    p, asg, pk, polys = setup(engine, A, L, F, k, lb, seed=0x5EED0019 + 3, worst=True)
    sh = plonk.Shape(k, A, L, F, lb)
    opk = prover.keygen(prover.Circuit(sh, asg.fixed, asg.copies, asg.advice))
    seed = b"\x42" * 32
    got = engine.prove(pk, polys, seed, E.ZK_TRANSCRIPT_EVM)
    assert got == prover.create_proof(opk, asg.advice, ChaCha20Rng(seed), "evm")
    for h in polys:
        h.free()
    engine.pk_free(pk)


def test_lookup_violation_is_reported(engine):
    A, L, F, k, lb = SHAPES["k19like"]
    p, asg, pk, polys = setup(engine, A, L, F, k, lb)
    # put an out-of-table value under q_lookup
    row = asg.fixed[asg.layout.fx_qlookup].index(1)
    col = list(asg.advice[0])
    col[row] = 1 << 40
    engine.upload_canonical(polys[0], asg.to_limbs(col))
    with pytest.raises(zk.ZkError) as e:
        engine.prove(pk, polys, bytes(32), E.ZK_TRANSCRIPT_EVM)
    assert e.value.code == -6  # ZK_EWITNESS
    for h in polys:
        h.free()
    engine.pk_free(pk)


@pytest.mark.parametrize("cfg", ["K17", "K19"])
def test_baseline_shapes_verify(engine, cfg):
    """BASELINE configs[1]/[2]: full-size proofs, sizes as published (K6), accepted by the pinned verifier."""
    p = getattr(zk.circuit, cfg)
    A, L, F, k, lb = p.num_advice, p.num_lookup_advice, p.num_fixed, p.degree, p.lookup_bits
    _, asg, pk, polys = setup(engine, A, L, F, k, lb)
    sh = plonk.Shape(k, A, L, F, lb)
    vk = product_vk(engine, pk, sh)
    want = {"K17": (1920, 2720), "K19": (960, 1536)}[cfg]
    pf = engine.prove(pk, polys, b"\x01" * 32, E.ZK_TRANSCRIPT_BLAKE2B)
    assert len(pf) == want[0]  # halo2-circuits/src/results/ecdsa_bench.csv:2,4
    assert plonk.verify(vk, pf, "blake2b")
    pe = engine.prove(pk, polys, b"\x01" * 32, E.ZK_TRANSCRIPT_EVM)
    assert len(pe) == want[1]
    assert plonk.verify(vk, pe, "evm")
    bad = bytearray(pe)
    bad[100] ^= 1
    assert not plonk.verify(vk, bytes(bad), "evm")
    for h in polys:
        h.free()
    engine.pk_free(pk)


# the remaining rows of halo2-circuits/src/configs/bench_ecdsa.config with the proof sizes the
# reference published for them (halo2-circuits/src/results/ecdsa_bench.csv:3,5-10)
# The k <= 13 rows are published 1 / 2 / 3 evaluations short of the full column shape: the only halo2
# mechanism that removes single elements is selector compression dropping the fixed column of a never-enabled
# selector, i.e. the circuit leaves its last 1 / 2 / 3 gate columns idle (last tuple entry).
BENCH_ROWS = [
    (18, 2, 1, 1, 17, 1344, 0),
    (16, 8, 2, 1, 15, 3552, 0),
    (15, 17, 3, 1, 14, 6560, 0),
    (14, 34, 6, 1, 13, 12704, 0),
    (13, 68, 12, 1, 12, 24960, 1),
    (12, 139, 24, 2, 11, 50496, 2),
    (11, 291, 53, 4, 10, 106496, 3),
    (20, 1, 1, 1, 19, 960, 0),  # not reference rows: above BASELINE's k=19 (same single-column shape); k=21 is
    (21, 1, 1, 1, 20, 960, 0),  # the size of the stress config and takes the 15-bit-window MSM path
]


@pytest.mark.parametrize("row", BENCH_ROWS, ids=lambda r: "k%d" % r[0])
def test_every_bench_config_row_proves(engine, row):
    """Full-size proofs of every other bench_ecdsa.config row: published size (K6) and accepted by the pinned
    verifier, with both multi-open schemes."""
    k, A, L, F, lb, size, idle = row
    _, asg, pk, polys = setup(engine, A, L, F, k, lb, idle=idle)
    sh = plonk.Shape(k, A, L, F, lb, idle)
    vk = product_vk(engine, pk, sh)
    pf = engine.prove(pk, polys, b"\x02" * 32, E.ZK_TRANSCRIPT_BLAKE2B)
    assert len(pf) == size == engine.proof_size(pk, E.ZK_TRANSCRIPT_BLAKE2B)
    assert plonk.verify(vk, pf, "blake2b")
    pe = engine.prove(pk, polys, b"\x02" * 32, E.ZK_TRANSCRIPT_EVM)
    assert len(pe) == engine.proof_size(pk, E.ZK_TRANSCRIPT_EVM)
    assert plonk.verify(vk, pe, "evm")
    bad = bytearray(pf)
    bad[len(bad) // 2] ^= 1
    assert not plonk.verify(vk, bytes(bad), "blake2b")
    for h in polys:
        h.free()
    engine.pk_free(pk)


def test_one_key_many_witnesses(engine):
    """The proving key is witness-independent (as the reference's pk is shared by every request,
    proving-server/src/main.rs:49-63): three jobs, one zk_keygen, all proofs byte-equal to the oracle."""
    A, L, F, k, lb = SHAPES["k19like"]
    p = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb)
    sh = plonk.Shape(k, A, L, F, lb)
    engine.srs_setup(k)
    asgs = [zk.circuit.synthesize(p, zk.batch.job_seed(i)) for i in range(3)]
    assert all(a.fixed == asgs[0].fixed and a.copies == asgs[0].copies for a in asgs)
    fixed = np.stack([asgs[0].to_limbs(c) for c in asgs[0].fixed])
    pk = engine.keygen(p, fixed, asgs[0].copies)
    opk = prover.keygen(prover.Circuit(sh, asgs[0].fixed, asgs[0].copies, asgs[0].advice))
    for i, asg in enumerate(asgs):
        h = engine.poly(1 << k)
        engine.upload_canonical(h, asg.to_limbs(asg.advice[0]))
        seed = i.to_bytes(32, "little")
        got = engine.prove(pk, [h], seed, E.ZK_TRANSCRIPT_EVM)
        assert got == prover.create_proof(opk, asg.advice, ChaCha20Rng(seed), "evm")
        h.free()
    engine.pk_free(pk)


def es256_request(seed=1):
    """A valid ES256 request in the server's encoding (five 32-byte little-endian fields)."""
    import random
    from webauthn_halo2_amd import ecdsa_p256 as api
    rng = random.Random(seed)
    d, kk, z = (rng.randrange(1, api._N) for _ in range(3))
    q = api._p256_mul(d, api._G)
    r = api._p256_mul(kk, api._G)[0] % api._N
    s = pow(kk, -1, api._N) * (z + r * d) % api._N
    le = lambda v: v.to_bytes(32, "little")
    return [le(q[0]), le(q[1]), le(r), le(s), le(z)]  # pubkey_x, pubkey_y, r, s, msg_hash


def test_reference_api_mirror(engine, tmp_path):
    """download_keys + the request-shaped provers (reference ecdsa_p256.rs:256-427) at the server's default
    degree 17 (proving-server/src/main.rs:17): sizes as published, accepted by the pinned verifier, same
    request -> same witness, error behaviour on bad input; an INVALID signature is refused (the
    synthetic stand-in must never turn arbitrary bytes into a verifying proof); the JSON-in / hex-out
    contract of the server (main.rs:39-79) on top."""
    import json
    from webauthn_halo2_amd import ecdsa_p256 as api, proving_server as srv

    api.shutdown()
    pkp, vkp = str(tmp_path / "proving_key.pk"), str(tmp_path / "verifying_key.vk")
    api.download_keys(17, pkp, vkp)
    eng = api._STATE[0]["eng"]
    req = es256_request()
    assert not hasattr(api, "generate_proof") and not hasattr(api, "generate_proof_evm")
    with pytest.raises(FileNotFoundError):
        api.generate_proof_synthetic(*req, str(tmp_path / "missing.pk"), 17)
    with pytest.raises(ValueError):
        api.generate_proof_synthetic(b"short", *req[1:], pkp, 17)
    with pytest.raises(ValueError):
        api.generate_proof_synthetic(*[bytes([i]) * 32 for i in range(5)], pkp, 17)   # arbitrary bytes: not a signature
    bad = list(req)
    bad[3] = (int.from_bytes(req[3], "little") ^ 2).to_bytes(32, "little")            # one bit of s flipped
    with pytest.raises(ValueError):
        api.generate_proof_evm_synthetic(*bad, pkp, 17)
    pf = api.generate_proof_synthetic(*req, pkp, 17, rng_seed=bytes(32))
    pe = api.generate_proof_evm_synthetic(*req, pkp, 17, rng_seed=bytes(32))
    assert (len(pf), len(pe)) == (1920, 2720)
    assert pe == api.generate_proof_evm_synthetic(*req, pkp, 17, rng_seed=bytes(32))
    assert pe != api.generate_proof_evm_synthetic(*req, pkp, 17)  # OsRng-style fresh randomness
    sh = plonk.Shape(17, 4, 1, 1, 16)
    p, pk = api._STATE[0]["keys"][pkp]
    vk = product_vk(eng, pk, sh)
    assert plonk.verify(vk, pf, "blake2b") and plonk.verify(vk, pe, "evm")
    # the verifying-key file is the VerifyingKey::write image (RawBytes, ecdsa_p256.rs:266-270): header, 6 + 6 commitments
    # (fixed columns in halo2's column order), the 4 selector bit vectors — what the oracle's serialiser gives for this key
    from zkoracle import serde as oserde
    asg0 = zk.circuit.synthesize(p, 0)
    want_vk = oserde.vk_bytes(sh, vk.fixed_commitments, vk.permutation_commitments, oserde.selectors_of(sh, asg0.fixed), oserde.RAW_BYTES)
    assert os.path.getsize(vkp) == 8 + (6 + 6) * 64 + 4 * (1 << 17) // 8 and open(vkp, "rb").read() == want_vk
    # /setup again under the same name: the resident key is replaced (the old one freed), not leaked
    api.download_keys(17, pkp, None)
    assert len(api._STATE[0]["keys"]) == 1
    p, pk = api._STATE[0]["keys"][pkp]
    # the engine's real input: advice columns handed over by the host
    asg = zk.circuit.synthesize(p, api._witness_seed(*req))
    assert api.create_proof_from_advice([asg.to_limbs(c) for c in asg.advice], pkp, 17, E.ZK_TRANSCRIPT_EVM, rng_seed=bytes(32)) == pe
    # JSON in, hex out
    body = {"pubkey_x": list(req[0]), "pubkey_y": list(req[1]), "r": list(req[2]), "s": list(req[3]), "msghash": list(req[4]),
            "proving_key_path": pkp}
    assert srv.prove_evm(json.dumps(body), rng_seed=bytes(32)) == pe.hex()
    assert bytes.fromhex(srv.prove(body, rng_seed=bytes(32))) == pf
    outs = srv.prove_batch([body, dict(body, r=[0] * 32), body], evm=True, devices=(0,))
    assert isinstance(outs[1], ValueError) and len(outs[0]) == len(outs[2]) == 2 * 2720
    assert plonk.verify(vk, bytes.fromhex(outs[2]), "evm")
    for brk in (dict(body, r=list(req[2])[:31]), dict(body, s=[256] + list(req[3])[1:]), {k: v for k, v in body.items() if k != "msghash"}):
        with pytest.raises(ValueError):
            srv.prove_evm(brk)
    # requests in flight side by side (ecdsa_p256.PIPELINES_PER_DEVICE pipelines on the device): six concurrent requests with
    # fixed blinding seeds give the bytes of the lone ones
    import threading
    conc = [None] * 6
    def one(i):
        conc[i] = api.generate_proof_evm_synthetic(*req, pkp, 17, rng_seed=bytes(32))
    ths = [threading.Thread(target=one, args=(i,)) for i in range(6)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert conc == [pe] * 6 and len(api._STATE[0]["extra"]) == api.PIPELINES_PER_DEVICE - 1
    api.shutdown()
    assert not api._STATE


def test_batch_of_jobs_equals_lone_proofs():
    """BASELINE config 4 in small: a batch of 10 independent jobs (witness seeds 0x5eed0019 + i) drained by two
    pipelines (own zk_ctx + host thread each, one shared device here) gives, job by job, the bytes a lone
    pipeline produces for that job — and those are the oracle's bytes."""
    from webauthn_halo2_amd import batch
    p = zk.circuit.CircuitParams(degree=9, num_advice=2, num_lookup_advice=1, num_fixed=1, lookup_bits=7)
    jobs = list(range(10))
    wit = batch.synthesize_jobs(p, jobs, processes=2)
    fixed, copies = batch.structure(p)
    pipes = [batch.Pipeline(0, p, fixed, copies, deterministic_seeds=True) for _ in range(2)]
    for q, pl in enumerate(pipes):
        for j in jobs[q::2]:
            pl.load(j, wit[j])
    got = batch.run(pipes, jobs, E.ZK_TRANSCRIPT_EVM)
    assert sorted(got) == jobs and len(set(got.values())) == len(jobs)
    lone = batch.Pipeline(0, p, fixed, copies, deterministic_seeds=True)
    for j in jobs:
        lone.load(j, wit[j])
        assert lone.prove(j, E.ZK_TRANSCRIPT_EVM) == got[j]
    # two of them against the oracle prover (same witness generator, same RNG stream)
    sh = plonk.Shape(9, 2, 1, 1, 7)
    asg0 = zk.circuit.synthesize(p, 0)
    opk = prover.keygen(prover.Circuit(sh, asg0.fixed, asg0.copies, asg0.advice))
    for j in (0, 7):
        asg = zk.circuit.synthesize(p, batch.job_seed(j))
        assert got[j] == prover.create_proof(opk, asg.advice, ChaCha20Rng(batch.job_rng_seed(j)), "evm")
    for pl in pipes + [lone]:
        pl.close()


def test_concurrent_pipelines_are_deterministic():
    """bench.py runs two proof pipelines (own context + host thread each) on one GPU: concurrent proofs of the
    same witness and seed must be the bytes a lone pipeline produces (k = 10: batched table path)."""
    import threading
    A, L, F, k, lb = 3, 2, 1, 10, 8
    p = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb)
    asg = zk.circuit.synthesize(p, 0x5EED0019)
    fixed = np.stack([asg.to_limbs(c) for c in asg.fixed])
    seed = b"\x33" * 32

    def pipeline(out, reps):
        eng = zk.Engine(0)
        eng.srs_setup(k)
        pk = eng.keygen(p, fixed, asg.copies)
        polys = []
        for col in asg.advice:
            h = eng.poly(1 << k)
            eng.upload_canonical(h, asg.to_limbs(col))
            polys.append(h)
        for _ in range(reps):
            out.append(eng.prove(pk, polys, seed, E.ZK_TRANSCRIPT_EVM))
            out.append(eng.prove(pk, polys, seed, E.ZK_TRANSCRIPT_BLAKE2B))
        eng.close()

    ref = []
    pipeline(ref, 1)
    outs = [[], []]
    ths = [threading.Thread(target=pipeline, args=(o, 6)) for o in outs]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for o in outs:
        assert len(o) == 12
        assert all(o[i] == ref[i % 2] for i in range(12))


@pytest.mark.parametrize("name", ["k19like", "k17like", "idle", "manycols"])
def test_quotient_entry_point_matches_oracle_h(engine, name):
    """zk_quotient (Evaluator::evaluate_h + divide_by_vanishing_poly for a host that drives the phases itself): fed the
    oracle prover's own intermediate columns (blinded advice, permutation products, permuted lookup columns — its `trace`)
    and challenges, the device quotient's coefficients equal the oracle's h(X), and the undivided numerator equals
    h(X) (X^n - 1) on the coset."""
    from zkoracle import field as F
    shp = SHAPES[name]
    A, L, Fx, k, lb = shp[:5]
    idle = shp[5] if len(shp) > 5 else 0
    p, asg, pk, polys = setup(engine, A, L, Fx, k, lb, idle=idle)
    sh = plonk.Shape(k, A, L, Fx, lb, idle)
    opk = prover.keygen(prover.Circuit(sh, asg.fixed, asg.copies, asg.advice))
    tr = {}
    prover.create_proof(opk, asg.advice, ChaCha20Rng(b"\x09" * 32), "evm", trace=tr)
    n, N = sh.n, 4 * sh.n
    shape = engine.pk_shape(pk)
    assert (shape["k"], shape["ext_k"], shape["n_advice"], shape["n_chunks"], shape["n_lookups"], shape["n_h"]) == \
           (k, k + 2, sh.n_adv, sh.n_chunks, sh.n_lookups, sh.n_h)

    def ext_of(values):
        v = engine.poly(n, cops.fr_mont(values))
        engine.lagrange_to_coeff(v)
        e = engine.poly(N)
        engine.coeff_to_extended(v, e)
        v.free()
        return e

    adv = [ext_of(c) for c in tr["adv"]]
    zs = [ext_of(z) for z in tr["zs"]]
    lks = [(ext_of(d["ap"]), ext_of(d["sp"]), ext_of(d["z"])) for d in tr["lk"]]
    out = engine.poly(N)
    ch = [cops.fr_mont([tr[c]])[0] for c in ("beta", "gamma", "y")]
    engine.quotient(pk, adv, zs, lks, *ch, out)
    engine.extended_to_coeff(out, N)
    assert cops.fr_ints(engine.download(out)) == tr["h_coeff"]
    # the bare numerator (divide = 0): h(X) * (X^n - 1) on the coset
    engine.quotient(pk, adv, zs, lks, *ch, out, divide=False)
    num = cops.fr_ints(engine.download(out, 8))
    hx = engine.poly(N, cops.fr_mont(tr["h_coeff"]))
    he = engine.poly(N)
    engine.coeff_to_extended(hx, he)
    hvals = cops.fr_ints(engine.download(he, 8))
    wext = F.omega(k + 2)
    for i in range(8):
        x = F.ZETA * pow(wext, i, F.R) % F.R
        assert num[i] == hvals[i] * (pow(x, n, F.R) - 1) % F.R
    with pytest.raises(zk.ZkError):
        engine.quotient(pk, adv[:-1] if len(adv) > 1 else [], zs, lks, *ch, out)   # wrong operand count
    with pytest.raises(zk.ZkError):
        engine.quotient(pk, adv, zs, lks, *ch, adv[0])                               # output aliases an input
    for h in adv + zs + [x for t in lks for x in t] + [out, hx, he] + polys:
        h.free()
    engine.pk_free(pk)


@pytest.mark.parametrize("name", ["k17like", "manycols"])
def test_quotient_rows_on_adversarial_cosets(engine, name):
    """The quotient kernel computes lazily on 29-bit limbs (value bounds tracked per expression, quotient.hip): its rows must
    equal the oracle's for ANY operand values, not only those of a satisfied circuit — cosets made of the largest stored
    words (p - 1), zeros, ones, alternations and random elements, with and without the division by X^n - 1."""
    from zkoracle import fastprover as fp, field as F
    A, L, Fx, k, lb = SHAPES[name][:5]
    p, asg, pk, polys = setup(engine, A, L, Fx, k, lb)
    sh = plonk.Shape(k, A, L, Fx, lb, 0)
    opk = fp.keygen(sh, asg.fixed, asg.copies)
    N = 4 * sh.n
    rng = np.random.default_rng(7)
    pm1 = np.array([(F.R - 1) >> (64 * i) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)  # stored words of the largest element

    def make(kind):
        if kind == "max":
            return np.tile(pm1, (N, 1))
        if kind == "zero":
            return np.zeros((N, 4), dtype=np.uint64)
        if kind == "one":
            return cops.fr_mont([1] * N)
        if kind == "alt":
            a = np.tile(pm1, (N, 1))
            a[::2] = 0
            return a
        v = np.frombuffer(rng.bytes(N * 32), dtype=np.uint64).reshape(N, 4).copy()
        v[:, 3] &= 0x0FFFFFFFFFFFFFFF
        return cops.fr_mont(cops.fr_ints(v))  # canonical random elements

    kinds = ["max", "zero", "one", "alt", "rand"]
    n_adv, n_z, n_lk = sh.n_adv, sh.n_chunks, sh.n_lookups
    for round_ in range(3):
        pick = lambda: make(kinds[int(rng.integers(len(kinds)))] if round_ else "max")
        adv_e = [pick() for _ in range(n_adv)]
        z_e = [pick() for _ in range(n_z)]
        lk_e = [(pick(), pick(), pick()) for _ in range(n_lk)]
        ch = [F.R - 1, F.R - 1, F.R - 1] if round_ == 0 else [int.from_bytes(rng.bytes(31), "little") for _ in range(3)]
        adv = [engine.poly(N, a) for a in adv_e]
        zs = [engine.poly(N, a) for a in z_e]
        lks = [tuple(engine.poly(N, a) for a in t) for t in lk_e]
        out = engine.poly(N)
        chm = [cops.fr_mont([c])[0] for c in ch]
        for divide in (True, False):
            engine.quotient(pk, adv, zs, lks, *chm, out, divide=divide)
            want = fp.evaluate_h(opk, adv_e, z_e, lk_e, *ch, divide=divide)
            assert np.array_equal(engine.download(out), want), (name, round_, divide)
        for h in adv + zs + [x for t in lks for x in t] + [out]:
            h.free()
    for h in polys:
        h.free()
    engine.pk_free(pk)


def _limbs(col):
    """ints -> (n, 4) uint64 canonical little-endian limbs"""
    a = np.zeros((len(col), 4), dtype=np.uint64)
    for i, v in enumerate(col):
        for q in range(4):
            a[i, q] = (v >> (64 * q)) & 0xFFFFFFFFFFFFFFFF
    return a


@pytest.mark.parametrize("shape,seed", [((8, 3, 2, 2, 5), 1234), ((8, 3, 2, 2, 5), 99), ((8, 1, 1, 1, 5), 1234), ((8, 1, 1, 1, 5), 7),
                                        ((9, 4, 1, 1, 6), 1234), ((10, 4, 1, 1, 7), 5)])
def test_adversarial_layout_matches_plain_python_oracle(engine, shape, seed):
    """zk_keygen / zk_prove on a column layout this repo's own generator (webauthn-halo2_amd/circuit.py) never
    produces — tests/adversarial_layout.py: overlapping gates on irregular rows, copy cycles of up to 40 cells over all
    permutation columns, constants on every row, lookup rows in one block, an all-zero gate column — byte-compared
    with the plain-Python oracle prover (both transcripts), verifying-key commitments included."""
    import adversarial_layout as adv

    k, A, L, F, lb = shape
    sh = plonk.Shape(k, A, L, F, lb)
    fixed, copies, advice = adv.build(sh, seed)
    adv.check(sh, fixed, copies, advice)
    opk = prover.keygen(prover.Circuit(sh, fixed, copies, advice))
    p = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb)
    engine.srs_setup(k)
    pk = engine.keygen(p, np.stack([_limbs(c) for c in fixed]), copies)
    vk = product_vk(engine, pk, sh)
    assert vk.fixed_commitments == opk.vk.fixed_commitments and vk.permutation_commitments == opk.vk.permutation_commitments
    assert vk.transcript_repr == opk.vk.transcript_repr
    polys = []
    for col in advice:
        h = engine.poly(1 << k)
        engine.upload_canonical(h, _limbs(col))
        polys.append(h)
    rseed = bytes([seed & 0xFF]) * 32
    for kind in ("evm", "blake2b"):
        got = engine.prove(pk, polys, rseed, KIND[kind])
        assert got == prover.create_proof(opk, advice, ChaCha20Rng(rseed), kind), (shape, seed, kind)
        assert plonk.verify(opk.vk, got, kind)
    for h in polys:
        h.free()
    engine.pk_free(pk)


@pytest.mark.parametrize("name", ["k17like", "k19like", "wide", "k10batched", "idle"])
def test_transcript_repr_is_halo2s_pinned_vk_hash(engine, name):
    """zk_keygen stamps `transcript_repr` as halo2 does: the Blake2b hash of the pinned verifying key's Debug rendering
    (csrc/vkrepr.h).  The oracle's restatement of that rendering (zkoracle/vkrepr.py) reproduces the reference's k = 17
    literal (tests/test_oracle_kat.py, P256Verifier.yul:34); here the device-side rendering gives the same digest as the
    oracle's for the key it has just made."""
    from zkoracle import vkrepr

    A, L, F, k, lb, idle = (SHAPES[name] + (0,))[:6]  # "idle": never-enabled gate columns — combined selectors in the rendering
    p, asg, pk, polys = setup(engine, A, L, F, k, lb, idle=idle)
    sh = plonk.Shape(k, A, L, F, lb, idle)
    vk = product_vk(engine, pk, sh)
    assert vk.transcript_repr == vkrepr.transcript_repr(sh, vk.fixed_commitments, vk.permutation_commitments)
    for h in polys:
        h.free()
    engine.pk_free(pk)


def test_k19_batch_of_8_jobs_equals_committed_oracle_proofs():
    """BASELINE configs[3] at its size: the first 8 jobs of the k = 19 batch workload (bench.py's timed steps) drained
    by two resident pipelines through batch.run — 960-byte Blake2b + SHPLONK proofs, byte for byte the proofs the
    oracle's CPU prover made for the same witness seeds and blinding streams (tests/golden/batch_k19_proofs.json)."""
    import hashlib
    import json

    from webauthn_halo2_amd import batch

    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "batch_k19_proofs.json")))
    jobs = sorted(int(j) for j in fx["jobs"])
    assert len(jobs) >= 8 and fx["degree"] == 19
    p = zk.circuit.K19
    wit = batch.synthesize_jobs(p, jobs)
    fixed, copies = batch.structure(p)
    # two pipelines (each MSM pass's reduction tail on its context's side stream), then bench.py's four (three or more proofs
    # in flight: the tails follow on the main streams — engine.hip ctx_msm_begin_batch)
    for npipe in (2, 4):
        pipes = [batch.Pipeline(0, p, fixed, copies, deterministic_seeds=True)]
        for _ in range(npipe - 1):
            pipes.append(batch.Pipeline(0, p, fixed, copies, deterministic_seeds=True, share_srs_with=pipes[0]))
        for q, pl in enumerate(pipes):
            for j in jobs[q::npipe]:
                pl.load(j, wit[j])
        got = batch.run(pipes, jobs, E.ZK_TRANSCRIPT_BLAKE2B)
        for pl in pipes[::-1]:
            pl.close()
        assert sorted(got) == jobs
        for j in jobs:
            assert len(got[j]) == 960
            assert hashlib.sha256(got[j]).hexdigest() == fx["jobs"][str(j)]["sha256"], (npipe, j)
            assert got[j].hex() == fx["jobs"][str(j)]["proof"], (npipe, j)


@pytest.mark.parametrize("mode", [1, 2], ids=["tails-on-side-stream", "tails-on-main-stream"])
def test_tail_stream_placement_does_not_change_proofs(mode):
    """ZK_OPT_MSM_TAIL_STREAM pins where the MSM reduction tails run (the default decides per pass from the proofs in flight
    on the device): the same bytes as the oracle's either way, at a shape whose commitments take the wide path (k = 10)."""
    A, L, F, k, lb = SHAPES["k10batched"][:5]
    eng = zk.Engine(0)
    eng.set_option(E.ZK_OPT_MSM_TAIL_STREAM, mode)
    p, asg, pk, polys = setup(eng, A, L, F, k, lb)
    sh = plonk.Shape(k, A, L, F, lb)
    opk = prover.keygen(prover.Circuit(sh, asg.fixed, asg.copies, asg.advice))
    seed = bytes([mode]) * 32
    for kind in ("blake2b", "evm"):
        assert eng.prove(pk, polys, seed, KIND[kind]) == prover.create_proof(opk, asg.advice, ChaCha20Rng(seed), kind)
    for h in polys:
        h.free()
    eng.pk_free(pk)
    eng.close()


def test_k19_batch_slice_through_four_pipelines_matches_oracle_digests():
    """BASELINE configs[3] as bench.py runs it on one GPU: a slice of the 256-job batch (jobs 40 .. 71: beyond the eight whose
    full bytes are committed) through FOUR pipelines — tails on the main streams — against the oracle's digests of the same jobs
    (tests/golden/batch_k19_sha256.json, tests/golden/make_batch_hashes.py)."""
    import hashlib
    import json

    from webauthn_halo2_amd import batch

    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "batch_k19_sha256.json")))["sha256"]
    jobs = [j for j in range(40, 72) if str(j) in want]
    assert len(jobs) >= 16
    p = zk.circuit.K19
    wit = batch.synthesize_jobs(p, jobs)
    fixed, copies = batch.structure(p)
    pipes = [batch.Pipeline(0, p, fixed, copies, deterministic_seeds=True)]
    for _ in range(3):
        pipes.append(batch.Pipeline(0, p, fixed, copies, deterministic_seeds=True, share_srs_with=pipes[0]))
    for q, pl in enumerate(pipes):
        for j in jobs[q::4]:
            pl.load(j, wit[j])
    got = batch.run(pipes, jobs, E.ZK_TRANSCRIPT_BLAKE2B)
    for pl in pipes[::-1]:
        pl.close()
    for j in jobs:
        assert len(got[j]) == 960 and hashlib.sha256(got[j]).hexdigest() == want[str(j)], j


def test_shared_srs_contexts():
    """zk_ctx_create_shared: a second context on the device uses the first one's resident SRS and window tables — same
    commitments, byte-identical proofs from its own key; the shared memory outlives the context that loaded it, and a
    context that loads another SRS for itself leaves the other one's view untouched."""
    A, L, F, k, lb = SHAPES["k10batched"][:5]
    p = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb)
    asg = zk.circuit.synthesize(p, 0x5EED0019)
    fixed = np.stack([asg.to_limbs(c) for c in asg.fixed])
    first = zk.Engine(0)
    first.srs_setup(k)
    second = zk.Engine(0, share_with=first)
    assert second.srs_msm_plan() == first.srs_msm_plan()
    a = np.frombuffer(np.random.default_rng(5).bytes(32 << k), dtype=np.uint64).reshape(-1, 4).copy()
    a[:, 3] &= 0x0FFFFFFFFFFFFFFF
    want = first.commit(first.poly(1 << k, a), 1)
    assert np.array_equal(second.commit(second.poly(1 << k, a), 1), want)
    proofs = []
    for eng in (first, second):
        pk = eng.keygen(p, fixed, asg.copies)
        polys = []
        for col in asg.advice:
            h = eng.poly(1 << k)
            eng.upload_canonical(h, asg.to_limbs(col))
            polys.append(h)
        proofs.append((eng, pk, polys, eng.prove(pk, polys, b"\x09" * 32, E.ZK_TRANSCRIPT_EVM)))
    assert proofs[0][3] == proofs[1][3]
    # the loader goes away first: the child keeps the SRS alive
    first.close()
    eng, pk, polys, pf = proofs[1]
    assert eng.prove(pk, polys, b"\x09" * 32, E.ZK_TRANSCRIPT_EVM) == pf
    assert np.array_equal(eng.commit(eng.poly(1 << k, a), 1), want)
    # a third context shares the second's; the second then loads another SRS for itself: the third still sees the old one
    third = zk.Engine(0, share_with=second)
    second.srs_setup(k, bytes([7]) * 32)
    assert not np.array_equal(second.commit(second.poly(1 << k, a), 1), want)
    assert np.array_equal(third.commit(third.poly(1 << k, a), 1), want)
    with pytest.raises(zk.ZkError):
        second.prove(pk, polys, b"\x09" * 32, E.ZK_TRANSCRIPT_EVM)  # the key was made under the SRS the context let go of
    second.close()
    third.close()


@pytest.mark.gpu
def test_poly_detach_attach_and_staged_loading():
    """zk_poly_detach / zk_poly_attach: a loader context uploads columns on its own stream and hands them to the proving
    context (no copy, new handle, the old one is gone); a proof from handed-over columns is byte-identical to one from columns loaded directly —
    also when a loader thread stages the next job while the pipeline proves the current one."""
    import threading
    from webauthn_halo2_amd import batch
    A, L, F, k, lb = SHAPES["k10batched"][:5]
    p = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb)
    asg = zk.circuit.synthesize(p, 0x5EED0019)
    fixed = np.stack([asg.to_limbs(c) for c in asg.fixed])
    pl = batch.Pipeline(0, p, fixed, asg.copies, deterministic_seeds=True)
    wit = {}
    for j in range(6):
        cols = [asg.to_limbs(c).copy() for c in asg.advice]
        wit[j] = cols
    # jobs differ through their seeds (job_rng_seed) — and in one cell-free way: identical columns, distinct blinding
    want = {}
    for j in wit:
        pl.load(j, wit[j])
        want[j] = pl.prove(j, E.ZK_TRANSCRIPT_BLAKE2B)
        pl.unload(j)
    # a detached vector is unknown to the loader, attaches once, and only on its device
    ld = pl.loader()
    h = ld.poly(1 << k)
    ld.upload_canonical(h, wit[0][0])
    old = h.h
    d = ld.poly_detach(h)
    assert h.h == 0
    assert ld.L.zk_poly_free(ld.ctx, old) != 0
    got = pl.eng.poly_attach(d)
    with pytest.raises(zk.ZkError):
        pl.eng.poly_attach(d)
    direct = pl.eng.poly(1 << k)
    pl.eng.upload_canonical(direct, wit[0][0])
    assert np.array_equal(pl.eng.download(got), pl.eng.download(direct))
    got.free()
    direct.free()
    # a staged column nobody adopts is discarded (zk_poly_discard): its token dies with it; a failed stage() discards the
    # columns it had already detached
    staged = pl.stage(wit[1])
    pl.discard(staged)
    with pytest.raises(zk.ZkError):
        pl.eng.poly_attach(staged[0])
    with pytest.raises(zk.ZkError):
        ld.poly_discard(staged[0])
    with pytest.raises(Exception):
        pl.stage([wit[1][0], np.zeros(((1 << k) + 1, 4), dtype=np.uint64)])  # the second upload is refused (too long)
    pl.adopt(0, pl.stage(wit[0]))
    assert pl.prove(0, E.ZK_TRANSCRIPT_BLAKE2B) == want[0]
    # overlapped: a thread stages job i+1 while job i is proved
    import queue
    q = queue.Queue(maxsize=2)

    def load():
        for j in wit:
            q.put((j, pl.stage(wit[j])))

    t = threading.Thread(target=load)
    t.start()
    for _ in wit:
        j, polys = q.get()
        pl.adopt(j, polys)
        assert pl.prove(j, E.ZK_TRANSCRIPT_BLAKE2B) == want[j]
    t.join()
    pl.close()


def _random_shapes(count, seed):
    """Circuit shapes drawn at random (fixed seed): small enough for the plain-Python oracle, spread over everything the
    engine branches on — one or many gate columns, 1 ... 8 lookups, idle gate columns, 6 <= k <= 9, 1 ... 3 constants columns."""
    import random
    pr = random.Random(seed)
    shapes = []
    while len(shapes) < count:
        k = pr.choice([6, 7, 7, 8, 8, 9])
        A = pr.choice([1, 2, 3, 5, 9, 17, 33, 45]) if k <= 8 else pr.choice([1, 2, 4, 9])
        L = 1 if A == 1 else pr.choice([1, 2, 3, 8 if A >= 9 else 2])
        F = pr.choice([1, 1, 2, 3])
        lb = pr.randrange(3, k)            # lookup table of 2^lb rows inside the usable rows
        idle = pr.choice([0, 0, 1, 2]) if A >= 3 else 0
        if 2 * idle > A:   # more never-enabled selectors than used ones would pair up in columns of their own: not modelled
            idle = A // 2
        shapes.append((A, L, F, k, lb, idle))
    return shapes


@pytest.mark.gpu
@pytest.mark.parametrize("shape", _random_shapes(10, 0x5EED0305), ids=lambda s: "A%dL%dF%dk%dlb%di%d" % s)
def test_random_shapes_byte_identical_to_oracle(engine, shape):
    """Proof bytes of randomly drawn circuit shapes (both transcripts) against the plain-Python oracle: every structural
    branch of the prover — column-batched passes, lanes-per-row quotient, argument-list combinations, chunk counts,
    selector compression with idle gates — under shapes nobody tuned it for."""
    A, L, F, k, lb, idle = shape
    p, asg, pk, polys = setup(engine, A, L, F, k, lb, seed=0x5EED0000 + 131 * A + k, idle=idle)
    sh = plonk.Shape(k, A, L, F, lb, idle)
    opk = prover.keygen(prover.Circuit(sh, asg.fixed, asg.copies, asg.advice))
    vk = product_vk(engine, pk, sh)
    assert vk.fixed_commitments == opk.vk.fixed_commitments
    assert vk.permutation_commitments == opk.vk.permutation_commitments
    assert vk.transcript_repr == opk.vk.transcript_repr
    seed = bytes([k, A & 255, L, F]) * 8
    for kind in ("blake2b", "evm"):
        got = engine.prove(pk, polys, seed, KIND[kind])
        want = prover.create_proof(opk, asg.advice, ChaCha20Rng(seed), kind)
        assert got == want, (shape, kind)
        assert plonk.verify(vk, got, kind)
    for h in polys:
        h.free()
    engine.pk_free(pk)
