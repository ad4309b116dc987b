"""The three-coset route of zk_prove (poly.hip "three cosets", ZK_OPT_QUOTIENT_DOMAIN): circuits whose quotient has three pieces
(two or more advice columns: deg h < 3n) take h over three of the extended domain's four cosets — three n-point transforms per
column, 3n quotient rows, a 3 x 3 solve per coefficient.  The pieces are the unique coefficients of h, so the bytes must equal
those of the whole-domain route (option = 1) and the oracle's (halo2's route).  The default takes the route from k = 16 (the
full-size parity tests and bench.py's k = 17 leg run it against the oracle's committed proofs / digests); option = 2 forces it
on the small shapes here, each of which is also proven by the plain-Python oracle."""
import numpy as np
import pytest

import webauthn_halo2_amd as zk
from webauthn_halo2_amd import engine as E

import test_gpu_prover as t

pytestmark = pytest.mark.gpu

# A, L, F, k, lookup_bits, idle: k below / at / above one NTT tile and one pass, plain and sliced quotient kernels, several
# chunks and lookups, combined selectors
SHAPES = [(2, 1, 1, 5, 3, 0), (2, 1, 1, 8, 5, 0), (4, 1, 1, 10, 7, 0), (4, 2, 2, 9, 6, 1), (17, 3, 1, 8, 5, 0), (3, 1, 1, 13, 10, 0),
          (44, 6, 2, 8, 6, 0)]


@pytest.mark.parametrize("shape", SHAPES)
def test_three_cosets_and_whole_domain_give_the_same_bytes(shape):
    A, L, F, k, lb, idle = shape
    seed = bytes([7 * k + A]) * 32
    out = {}
    for full in (0, 1):
        eng = zk.Engine(0)
        eng.set_option(E.ZK_OPT_QUOTIENT_DOMAIN, 1 if full else 2)  # (2: also below the auto rule's k = 16)
        p, asg, pk, polys = t.setup(eng, A, L, F, k, lb, seed=0x5EED3000 + k, idle=idle)
        assert eng.pk_shape(pk)["n_h"] == 3  # three pieces: the route applies
        out[full] = [eng.prove(pk, polys, seed, t.KIND[kind]) for kind in ("blake2b", "evm")]
        out[full].append(eng.prove(pk, polys, seed, t.KIND["blake2b"]))  # again: the key's coset-major copies are reused
        if not full and k <= 9:  # the oracle's own bytes (plain Python: small shapes only)
            sh = t.plonk.Shape(k, A, L, F, lb, idle)
            opk = t.prover.keygen(t.prover.Circuit(sh, asg.fixed, asg.copies, asg.advice))
            assert out[full][1] == t.prover.create_proof(opk, asg.advice, t.ChaCha20Rng(seed), "evm")
        eng.close()
    assert out[0] == out[1]
    assert out[0][0] == out[0][2]


def test_single_column_shape_keeps_the_whole_domain():
    """One advice column: the lookup input q_lookup * a makes the circuit's degree 5 — four pieces, every coset needed."""
    eng = zk.Engine(0)
    p, asg, pk, polys = t.setup(eng, 1, 1, 1, 8, 5)
    assert eng.pk_shape(pk)["n_h"] == 4
    a = eng.prove(pk, polys, bytes(32), t.KIND["blake2b"])
    eng.set_option(E.ZK_OPT_QUOTIENT_DOMAIN, 2)  # forced: still not applicable
    assert eng.prove(pk, polys, bytes(32), t.KIND["blake2b"]) == a
    eng.close()


def test_full_size_k17_server_shape_both_routes():
    """The proving server's k = 17 shape (main.rs:17) at full size: both routes, both transcripts, same bytes; the audit ledger
    watches the three-coset route's enqueues."""
    out = {}
    for full in (0, 1):
        eng = zk.Engine(0)
        eng.set_option(E.ZK_OPT_QUOTIENT_DOMAIN, full)
        if not full:
            eng.set_option(E.ZK_OPT_STREAM_AUDIT, 1)
        p, asg, pk, polys = t.setup(eng, 4, 1, 1, 17, 16)
        out[full] = [eng.prove(pk, polys, bytes(range(32)), t.KIND[kind]) for kind in ("blake2b", "evm")]
        if not full:
            checks, violations, msg = eng.audit_report()
            assert checks > 0 and violations == 0, msg
        eng.close()
    assert out[0] == out[1]
    assert len(out[0][0]) == 1920 and len(out[0][1]) == 2720


@pytest.mark.parametrize("shape", [(2, 1, 1, 8, 5, 0), (4, 2, 2, 9, 6, 1)])
def test_lock_step_batches_take_the_route_too(shape):
    """zk_prove_batch: the members read the key's coset-major copies through their own records — made before and after the
    members exist — and every proof equals its lone proof on either route."""
    import test_gpu_prove_batch as tb

    seeds = [0x5EED3100 + i for i in range(4)]
    rng = [bytes([40 + i]) * 32 for i in range(4)]
    want = None
    for order in ("batch first", "lone first", "whole domain"):
        eng = zk.Engine(0)
        eng.set_option(E.ZK_OPT_QUOTIENT_DOMAIN, 1 if order == "whole domain" else 2)
        pk, sets, asgs, opk = tb._setup(eng, shape, seeds)
        if order == "lone first":
            lone = [eng.prove(pk, sets[j], rng[j], t.KIND["evm"]) for j in range(4)]
            got = eng.prove_batch(pk, sets, rng, t.KIND["evm"])
        else:
            got = eng.prove_batch(pk, sets, rng, t.KIND["evm"])
            lone = [eng.prove(pk, sets[j], rng[j], t.KIND["evm"]) for j in range(4)]
        assert got == lone, order
        assert want is None or got == want, order
        want = got
        eng.close()


@pytest.mark.parametrize("shape,seed", [((8, 3, 2, 2, 5), 1234), ((9, 4, 1, 1, 6), 99), ((10, 4, 1, 1, 7), 5)])
def test_adversarial_layouts_on_three_cosets(shape, seed):
    """The layouts this repo's generator never produces (tests/adversarial_layout.py: overlapping gates on irregular rows, long copy
    cycles, constants on every row, an all-zero gate column) through the three-coset route, against the plain-Python oracle."""
    eng = zk.Engine(0)
    eng.set_option(E.ZK_OPT_QUOTIENT_DOMAIN, 2)
    try:
        t.test_adversarial_layout_matches_plain_python_oracle(eng, shape, seed)  # (the test's body: keygen, both transcripts, verify)
    finally:
        eng.close()
