"""The oracle's full-size CPU prover (oracle/zkoracle/fastprover.py: numpy + C operators) is pinned, byte for byte,
to the plain-Python restatement of create_proof (zkoracle.prover) — which is itself pinned by the verifier that
accepts the reference's golden proof.  The BASELINE-size golden proofs under tests/golden/ come from it."""
import numpy as np
import pytest

import webauthn_halo2_amd as zk
from zkoracle import cops, fastprover as fp, plonk, prover
from zkoracle.hashes import ChaCha20Rng

SHAPES = {
    "k19like": (1, 1, 1, 7, 6, 0),
    "k17like": (4, 1, 1, 7, 5, 0),
    "wide": (3, 2, 2, 8, 6, 0),
    "idle": (5, 2, 2, 7, 5, 2),
}


def make(name, seed=0x5EED0019, worst=False):
    A, L, F, k, lb, idle = SHAPES[name]
    p = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb, idle_gate_columns=idle)
    asg = zk.circuit.synthesize(p, seed, worst_case=worst)
    return plonk.Shape(k, A, L, F, lb, idle), asg


@pytest.mark.parametrize("name", list(SHAPES))
def test_fast_prover_equals_plain_python_prover(name):
    sh, asg = make(name)
    opk = prover.keygen(prover.Circuit(sh, asg.fixed, asg.copies, asg.advice))
    fpk = fp.keygen(sh, asg.fixed, asg.copies)
    assert fpk.vk.fixed_commitments == opk.vk.fixed_commitments
    assert fpk.vk.permutation_commitments == opk.vk.permutation_commitments
    assert fpk.vk.transcript_repr == opk.vk.transcript_repr
    for kind, scheme in (("evm", None), ("blake2b", None), ("evm", "shplonk"), ("blake2b", "gwc")):
        seed = bytes([len(name), len(kind)]) * 16
        want = prover.create_proof(opk, asg.advice, ChaCha20Rng(seed), kind, scheme)
        got = fp.create_proof(fpk, asg.advice, ChaCha20Rng(seed), kind, scheme)
        assert got == want, (name, kind, scheme)
        assert plonk.verify(opk.vk, got, kind, scheme)


def test_fast_prover_with_real_msms_gives_the_same_bytes():
    """commit mode "msm" (best_multiexp restatement over generated SRS bases, what cpu_baseline times) ==
    commit mode "tau" (the secret-key shortcut the fixtures use)."""
    sh, asg = make("k17like", seed=0x5EED0019 + 2, worst=True)
    cm = fp.Committer(sh.k, "msm")
    a = fp.create_proof(fp.keygen(sh, asg.fixed, asg.copies, cm), asg.advice, ChaCha20Rng(b"\x07" * 32), "blake2b", committer=cm)
    b = fp.create_proof(fp.keygen(sh, asg.fixed, asg.copies), asg.advice, ChaCha20Rng(b"\x07" * 32), "blake2b")
    assert a == b and cm.count > 0


def test_lookup_permutation_vectorised_equals_reference_form():
    sh, asg = make("k19like")
    n, usable = sh.n, sh.usable_rows
    inp = [asg.fixed[sh.fx_qlookup][i] * asg.advice[0][i] % plonk.R for i in range(n)]
    tab = asg.fixed[sh.fx_table]

    class Fixed:
        def __init__(self):
            self.i = 0

        def fr(self):
            self.i += 1
            return self.i

    a, s = prover.permute_expression_pair(inp, tab, usable, Fixed())
    blind = list(range(1, 8)), list(range(8, 15))
    fa, fs = fp.permute_expression_pair(fp.arr(inp), fp.arr(tab), usable, *blind)
    assert cops.fr_ints(fa) == a and cops.fr_ints(fs) == s
    bad = list(inp)
    bad[5] = 1 << sh.lookup_bits  # outside the table
    with pytest.raises(ValueError):
        fp.permute_expression_pair(fp.arr(bad), fp.arr(tab), usable, *blind)


def test_vector_operators_against_python_ints():
    rng = np.random.default_rng(3)
    R = plonk.R
    xs = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(300)]
    ys = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(300)]
    ys[7] = 0
    a, b = fp.arr(xs), fp.arr(ys)
    assert cops.fr_ints(fp.lin(a, 5, b, 7, 11)) == [(5 * x + 7 * y + 11) % R for x, y in zip(xs, ys)]
    assert cops.fr_ints(fp.mul(a, b)) == [x * y % R for x, y in zip(xs, ys)]
    assert cops.fr_ints(fp.batch_inv(b)) == [pow(y, -1, R) if y else 0 for y in ys]
    assert fp.dot(a, b) == sum(x * y for x, y in zip(xs, ys)) % R
    z = 0x1234567
    assert fp.eval_poly(a, z) == sum(x * pow(z, i, R) for i, x in enumerate(xs)) % R
    assert cops.fr_ints(fp.kate_division(a, z))[:-1] == prover.kate_division(xs, z)
    run = [3]
    for y in ys[:-1]:
        run.append(run[-1] * y % R)
    assert cops.fr_ints(fp.running_product(b, 3)) == run
    r = ChaCha20Rng(b"\x09" * 32)
    want = [r.fr() for _ in range(40)]
    assert cops.fr_ints(fp.chacha_fr(b"\x09" * 32, 0, 40)) == want
    assert cops.fr_ints(fp.chacha_fr(b"\x09" * 32, 17, 5)) == want[17:22]
