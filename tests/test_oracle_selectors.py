"""halo2's selector compression on the synthesized circuits (CPU): the closed form the oracle, the engine and the
`transcript_repr` rendering use (zkoracle/plonk.py Shape.gate_sel) is what the restated `compress_selectors::process`
(zkoracle/selectors.py) gives on the circuits' real selector activations — never-enabled gate columns included — and the
proof-size consequence is the reference's published one (halo2-circuits/src/results/ecdsa_bench.csv:8-10)."""
import pytest

import webauthn_halo2_amd as zk
from zkoracle import plonk, selectors, vkrepr


def raw_selectors(p, asg):
    """The circuit's selectors BEFORE compression, in halo2-lib's allocation order: one simple selector per gate column
    (never-enabled for the idle ones), then — single-column strategy — the complex q_lookup."""
    lay = asg.layout
    n = 1 << p.degree
    acts, degs = [], []
    for j in range(p.num_advice):
        col = lay.fx_sel[j]
        acts.append([False] * n if col is None else [v == 1 for v in asg.fixed[col]])
        degs.append(3)  # q * (a + b * c - d)
    if lay.single:
        acts.append([v == 1 for v in asg.fixed[lay.fx_qlookup]])
        degs.append(0)  # complex: occurs in the lookup only
    return acts, degs


@pytest.mark.parametrize("shape", [(1, 1, 1, 7, 6, 0), (4, 1, 1, 7, 5, 0), (5, 2, 2, 7, 5, 2), (4, 1, 1, 6, 4, 1), (9, 3, 1, 6, 4, 4)])
def test_gate_sel_is_what_compress_selectors_gives(shape):
    A, L, F, k, lb, idle = shape
    p = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb, idle_gate_columns=idle)
    asg = zk.circuit.synthesize(p, 0x5EED0019)
    sh = plonk.Shape(k, A, L, F, lb, idle)
    acts, degs = raw_selectors(p, asg)
    cols, asn = selectors.process(acts, degs, sh.degree)
    base = F + 1  # the selector columns follow the constants and the table (query order)
    assert len(cols) == sh.n_fix - base
    for j in range(A):
        c, root, members = asn[j]
        col, form = sh.gate_sel[j]
        assert col == base + c
        assert form == (plonk.SEL_PLAIN if members == 1 else plonk.SEL_FIRST_OF_TWO if root == 1 else plonk.SEL_SECOND_OF_TWO)
        assert members <= 2
    if sh.single:
        assert asn[A][0] + base == sh.fx_qlookup and asn[0][0] + base == sh.fx_sel[0]  # the complex selector's column comes first
    # the compressed columns ARE the fixed columns of the synthesized circuit (a combined column holds 1 where its used
    # member is enabled: the never-enabled member would write 2, never)
    for c, vals in enumerate(cols):
        assert vals == [int(v) for v in asg.fixed[base + c]]
    # the selector expression takes the value 1 exactly where the used selector is enabled, 0 elsewhere, in every form
    for j in range(A):
        col, form = sh.gate_sel[j]
        for r in (0, 1, 2, 3, 5, 8):
            want = 1 if (acts[j][r]) else 0
            assert plonk.selector_value(form, int(asg.fixed[col][r])) == want


def test_rendering_with_combined_selectors():
    sh0 = plonk.Shape(7, 5, 2, 2, 5, 0)
    sh2 = plonk.Shape(7, 5, 2, 2, 5, 2)
    pt = (1, 2)
    s0 = vkrepr.pinned_debug(sh0, [pt] * sh0.n_fix, [pt] * len(sh0.perm_cols))
    s2 = vkrepr.pinned_debug(sh2, [pt] * sh2.n_fix, [pt] * len(sh2.perm_cols))
    two = "Constant(0x%064x)" % 2
    one = "Constant(0x%064x)" % 1
    assert two not in s0 and one not in s0
    assert s2.count(two) == 2 and s2.count(one) == 2  # gates 0, 1 carry q (2 - q); gates 3, 4 (never enabled) q (1 - q)
    assert s2.count("Product(Product(Fixed {") == 4
    # same number of gates and selectors, two fixed columns (and queries) fewer
    assert s0.count("Sum(Sum(Advice") == s2.count("Sum(Sum(Advice") == 5
    assert "num_selectors: 5" in s0 and "num_selectors: 5" in s2
    assert "num_fixed_columns: %d" % sh0.n_fix in s0 and "num_fixed_columns: %d" % (sh0.n_fix - 2) in s2
    # gate 0's selector column is shared with gate 3 (the first never-enabled one)
    g0 = "Fixed { query_index: %d, column_index: %d, rotation: Rotation(0) }" % (sh2.gate_sel[0][0], vkrepr.halo2_fixed_column(sh2, sh2.gate_sel[0][0]))
    assert s2.count(g0) == 4  # twice in gate 0 (q and 2 - q), twice in gate 3 (q and 1 - q); the fixed-queries list prints columns


@pytest.mark.parametrize("row", [(13, 68, 12, 1, 12, 24960, 1), (12, 139, 24, 2, 11, 50496, 2), (11, 291, 53, 4, 10, 106496, 3)])
def test_published_sizes_of_the_rows_with_idle_columns(row):
    k, A, L, F, lb, size, idle = row
    sh = plonk.Shape(k, A, L, F, lb, idle)
    points = sh.n_points_before_multiopen() + 2  # SHPLONK: 2 opening commitments
    assert 32 * (points + sh.n_evals()) == size  # halo2-circuits/src/results/ecdsa_bench.csv:8-10
    full = plonk.Shape(k, A, L, F, lb, 0)
    assert 32 * (full.n_points_before_multiopen() + 2 + full.n_evals()) == size + 32 * idle
