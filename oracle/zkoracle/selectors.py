"""TEST INFRASTRUCTURE (oracle) — halo2's selector compression, restated.

halo2_proofs `plonk/circuit/compress_selectors.rs` `process` [RECALLED: the source is an un-vendored dependency of the
reference; called from `ConstraintSystem::compress_selectors`, reached from `keygen_vk` / `keygen_pk`, reference call sites
halo2-circuits/src/ecc/ecdsa_p256.rs:259-260]:
  * selectors of degree 0 (complex selectors and selectors that occur in no gate) get a fixed column each, first, in
    selector order; their expression is the column's query;
  * the simple ones are combined greedily: for selector i (not yet placed) the later selectors j join its combination in
    order as long as j is not enabled on a row where a member is (exclusion matrix) and
        max(d, deg_j - 1) + members + 1 <= max_degree     (d = the largest gate degree of the members minus 1),
    the scan stopping as soon as d + members == max_degree;
  * a combination of m selectors shares ONE fixed column q holding the member's number (1 .. m) on the rows where that
    member is enabled and 0 elsewhere; member r's expression is  q * prod_{s = 1..m, s != r} (s - q).
`Shape.gate_sel` (zkoracle/plonk.py) is the closed form of this for halo2-lib's constraint system (gate degree 3,
constraint-system degree 4: at most two members; used gate columns exclude each other, never-enabled ones exclude
nobody); tests/test_oracle_selectors.py runs this restatement on the synthesized circuits' real activations and compares.
PINNING: for num_idle_gate_columns = 0 the result is the one-column-per-selector layout whose rendering reproduces the
reference's `transcript_repr` literal (K3, tests/test_oracle_kat.py); combinations of two have no known answer in the
reference — only the proof sizes of the k <= 13 rows (ecdsa_bench.csv:8-10), which they explain."""


def process(activations, degrees, max_degree):
    """activations[s]: list of bools (rows); degrees[s]: the selector's largest gate degree (0: complex / in no gate).
    -> (columns, assignment): columns[c] = list of small ints (the fixed column's values), assignment[s] =
    (column index, root, members of the combination)."""
    ns = len(activations)
    n = len(activations[0]) if ns else 0
    columns, assignment = [], {}
    simple = []
    for s in range(ns):
        if degrees[s] == 0:
            assignment[s] = (len(columns), 1, 1)
            columns.append([1 if b else 0 for b in activations[s]])
        else:
            simple.append(s)
    rows = [frozenset(i for i, b in enumerate(activations[s]) if b) for s in range(ns)]
    conflict = lambda a, b: not rows[a].isdisjoint(rows[b])
    added = set()
    for pos, i in enumerate(simple):
        if i in added:
            continue
        added.add(i)
        assert degrees[i] <= max_degree
        d = degrees[i] - 1
        comb = [i]
        for j in simple[pos + 1:]:
            if d + len(comb) == max_degree:
                break
            if j in added or any(conflict(j, m) for m in comb):
                continue
            new_d = max(d, degrees[j] - 1)
            if new_d + len(comb) + 1 > max_degree:
                continue
            d = new_d
            comb.append(j)
            added.add(j)
        col = [0] * n
        for root, s in enumerate(comb, start=1):
            for r in rows[s]:
                col[r] = root
            assignment[s] = (len(columns), root, len(comb))
        columns.append(col)
    return columns, assignment
