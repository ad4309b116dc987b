"""Host-side circuit description and the synthetic same-shape witness.

The reference's `ECDSACircuit::synthesize` (halo2-circuits/src/ecc/ecdsa_p256.rs:117-206)
fills the advice columns through halo2-lib's secp256r1 chips; it stays on the
host and is out of the engine's scope (BASELINE.json north_star), and it cannot
be reproduced here (no Rust, halo2-lib not vendored — SURVEY.md §7 hard part vi).
What the engine consumes is its *result*: advice columns, fixed columns and copy
constraints of the column shape that `CircuitParams` selects
(halo2-circuits/src/configs/bench_ecdsa.config).  `synthesize()` builds a
satisfying assignment of exactly that shape (halo2-lib's vertical gate
q*(a + b*c - d), range-table lookup, copy constraints, constants column) from
the value mix SURVEY.md §8d prescribes.
"""
import json
import random
from dataclasses import dataclass

import numpy as np

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
BLINDING_FACTORS = 6


@dataclass
class CircuitParams:
    """Mirror of the reference's JSON config rows (ecdsa_p256.rs:44-55)."""
    strategy: str = "Simple"
    degree: int = 17
    num_advice: int = 4
    num_lookup_advice: int = 1
    num_fixed: int = 1
    lookup_bits: int = 16
    limb_bits: int = 88
    num_limbs: int = 3
    # not a key of the reference's JSON: how many of the gate columns (the last ones) the synthesized circuit
    # never enables.  halo2's selector compression gives those no fixed column, which is what the published
    # proof sizes of the k <= 13 rows imply (ecdsa_bench.csv:8-10 are 1/2/3 evaluations short of the full shape)
    idle_gate_columns: int = 0

    @staticmethod
    def from_json(line: str):
        return CircuitParams(**json.loads(line))


# the rows of halo2-circuits/src/configs/bench_ecdsa.config that BASELINE.json names
K19 = CircuitParams(degree=19, num_advice=1, num_lookup_advice=1, num_fixed=1, lookup_bits=18)
K17 = CircuitParams(degree=17, num_advice=4, num_lookup_advice=1, num_fixed=1, lookup_bits=16)


class Layout:
    """Column layout as a function of the config."""

    def __init__(self, p: CircuitParams):
        self.params = p
        A, L, F = p.num_advice, p.num_lookup_advice, p.num_fixed
        self.k = p.degree
        self.n = 1 << self.k
        self.single = A == 1
        self.n_gate = A
        self.n_lookup_cols = 0 if self.single else L
        self.n_adv = A + self.n_lookup_cols
        self.fx_table = F
        if self.single:
            # halo2's selector compression allocates the complex selector's fixed column before the gate selector's
            self.fx_qlookup = F + 1
            self.fx_sel = [F + 2]
            self.n_fix = F + 3
        else:
            U = p.idle_gate_columns
            self.fx_sel = [F + 1 + j for j in range(A - U)] + [None] * U
            self.fx_qlookup = None
            self.n_fix = F + 1 + A - U
        # permutation columns: constants, gate advice, lookup advice
        self.perm_cols = [("fixed", f) for f in range(F)] + [("advice", j) for j in range(self.n_adv)]
        self.usable_rows = self.n - (BLINDING_FACTORS + 1)

    def perm_index(self, kind, idx):
        return self.perm_cols.index((kind, idx))


class Assignment:
    """Result of synthesis: canonical integers, column-major."""

    def __init__(self, layout, fixed, copies, advice):
        self.layout = layout
        self.fixed = fixed      # [n_fix][n]
        self.copies = copies    # [((perm_col, row), (perm_col, row))]
        self.advice = advice    # [n_adv][n]

    @staticmethod
    def to_limbs(col):
        """list of ints -> (n, 4) uint64 canonical little-endian limbs."""
        b = b"".join(int(v).to_bytes(32, "little") for v in col)
        return np.frombuffer(b, dtype=np.uint64).reshape(-1, 4).copy()


STRUCT_SEED = 0xC1BC0019  # fixes the circuit (selectors, lookup flags, copy constraints); witnesses vary per job


def synthesize(p: CircuitParams, seed: int, worst_case: bool = False, struct_seed: int = STRUCT_SEED) -> Assignment:
    """Satisfying assignment of the config's shape.

    The circuit structure — which cells are range-checked, which are copies of earlier gate
    outputs or of constants — is drawn from `struct_seed` only, so every job shares one proving
    key, as every request shares the reference's pk (proving-server/src/main.rs:49-63).  The
    witness VALUES are drawn from `seed` (0x5eed0019 + job index).
    worst_case: every free value uniform in Fr (range-checked cells stay in range)."""
    lay = Layout(p)
    srng = random.Random(struct_seed)   # structure
    rng = random.Random(seed)           # values
    n, usable, lb = lay.n, lay.usable_rows, p.lookup_bits
    T = 1 << lb
    assert T < usable, "range table must fit in the usable rows"
    F = p.num_fixed
    fixed = [[0] * n for _ in range(lay.n_fix)]
    advice = [[0] * n for _ in range(lay.n_adv)]
    copies = []
    # constants column(s): a few small constants at the top
    n_const = min(64, usable)
    for f in range(F):
        for r in range(n_const):
            fixed[f][r] = (r * (f + 1)) % R
    # range table 0..2^lookup_bits-1, then zeros
    for r in range(T):
        fixed[lay.fx_table][r] = r

    def pick_class():
        u = srng.random()
        if u < 0.40:
            return "small"
        if u < 0.75:
            return "limb"
        if u < 0.90:
            return "full"
        return "zero"

    def value(cls):
        if cls == "small":
            return rng.randrange(T)
        if worst_case:
            return rng.randrange(R)
        if cls == "limb":
            return rng.randrange(1 << p.limb_bits)
        if cls == "full":
            return rng.randrange(R)
        return 0

    gates_per_col = usable // 4
    small_cells = []  # (gate advice col, row) range-checked cells
    d_cells = []      # (col, row) gate outputs available for copying (structure) ...
    d_vals = {}       # ... and their values (witness)
    for j in range(lay.n_gate):
        if lay.fx_sel[j] is None:
            continue  # idle gate column: nothing assigned
        col = advice[j]
        sel = fixed[lay.fx_sel[j]]
        for g in range(gates_per_col):
            r0 = 4 * g
            ca, cb, cc = pick_class(), pick_class(), pick_class()
            a, b, c = value(ca), value(cb), value(cc)
            u = srng.random()
            if d_cells and u < 0.5:
                # a := an earlier gate output (copy constraint)
                sc, sr = d_cells[srng.randrange(len(d_cells))]
                a, ca = d_vals[(sc, sr)], "copy"
                copies.append(((lay.perm_index("advice", j), r0), (lay.perm_index("advice", sc), sr)))
            elif u < 0.6:
                # b := a constant from the constants column
                f = srng.randrange(F)
                cr = srng.randrange(n_const)
                b, cb = fixed[f][cr], "const"
                copies.append(((lay.perm_index("fixed", f), cr), (lay.perm_index("advice", j), r0 + 1)))
            d = (a + b * c) % R
            col[r0], col[r0 + 1], col[r0 + 2], col[r0 + 3] = a, b, c, d
            sel[r0] = 1
            if ca == "small":
                small_cells.append((j, r0))
            if cc == "small":
                small_cells.append((j, r0 + 2))
            if len(d_cells) < 4096:
                d_cells.append((j, r0 + 3))
            else:
                old = srng.randrange(4096)
                d_vals.pop(d_cells[old], None)
                d_cells[old] = (j, r0 + 3)
            d_vals[(j, r0 + 3)] = d
    if lay.single:
        # the gate column itself is looked up under q_lookup
        ql = fixed[lay.fx_qlookup]
        for (_, row) in small_cells:
            ql[row] = 1
    else:
        # dedicated lookup columns: cell t holds a copy of a range-checked gate cell
        per = (len(small_cells) + lay.n_lookup_cols - 1) // max(lay.n_lookup_cols, 1)
        for l in range(lay.n_lookup_cols):
            cells = small_cells[l * per:(l + 1) * per][:usable]
            lc = advice[lay.n_gate + l]
            for t, (j, row) in enumerate(cells):
                lc[t] = advice[j][row]
                copies.append(((lay.perm_index("advice", lay.n_gate + l), t), (lay.perm_index("advice", j), row)))
    return Assignment(lay, fixed, copies, advice)
