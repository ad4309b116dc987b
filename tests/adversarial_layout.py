"""A second circuit-structure / witness generator for the halo2-lib column shape, written independently of
webauthn-halo2_amd/circuit.py (it shares no code with it and makes the opposite choice wherever that generator makes
one), for tests/test_gpu_prover.py::test_adversarial_layout_matches_plain_python_oracle:

  circuit.py                                   here
  selector on every 4th row, gates disjoint    selectors on irregular rows: overlapping gates (the output of one is an
                                               input of the next), dense bursts, long gaps, one column almost unused
  copy constraints are pairs                   copy CYCLES of 2 .. 40 cells across ALL permutation columns (constants,
                                               every gate column, the lookup columns), several cells of one column in a cycle
  64 constants at the top of the column        constants columns populated on every usable row
  range-checked cells spread over the column   lookup rows in ONE block (q_lookup block / non-zero block of a lookup column)
  every advice column busy                     one gate column all zero (its few gates are 0 + 0 * 0 = 0)

Returns the same objects the oracle prover and zk_keygen take: fixed columns in the engine's order (constants, table,
selectors), copies as ((perm_col, row), (perm_col, row)) pairs chaining each cycle, advice columns."""
import random

from zkoracle.field import R


def build(shape, seed, disjoint_selectors=False):
    """disjoint_selectors: leave the gate selectors as drawn — the almost-unused column's five rows then usually miss some
    other column's rows, a layout halo2 would compress differently (the refusal test)."""
    rng = random.Random(seed)
    n, usable, F, A = shape.n, shape.usable_rows, shape.num_fixed, shape.n_gate
    T = 1 << shape.lookup_bits
    assert T < usable
    fixed = [[0] * n for _ in range(shape.n_fix)]
    advice = [[0] * n for _ in range(shape.n_adv)]
    for f in range(F):  # constants on every usable row: small, large, zero, p - 1
        for r in range(usable):
            fixed[f][r] = rng.choice((rng.randrange(T), rng.randrange(R), 0, R - 1, rng.randrange(1 << 88)))
    for r in range(T):
        fixed[shape.fx_table][r] = r
    zero_col = A - 1 if A >= 3 else None

    # ---- lookup block
    blk_len = max(8, usable // 5)
    blk0 = rng.randrange(8, usable - blk_len - 8)
    block = range(blk0, blk0 + blk_len)

    common_row = 2
    # ---- selectors: irregular rows, overlapping gates; no gate OUTPUT inside the lookup block of a looked-up gate column
    sel_rows = []
    for j in range(A):
        rows = set()
        if j == zero_col:
            rows = set(rng.sample(range(usable - 4), 5))
        else:
            r = rng.randrange(3)
            while r < usable - 4:
                mode = rng.random()
                if mode < 0.15:      # burst of overlapping gates on consecutive rows
                    for q in range(rng.randrange(3, 12)):
                        if r + q < usable - 4:
                            rows.add(r + q)
                    r += 12
                elif mode < 0.25:    # long gap
                    r += rng.randrange(20, 60)
                else:
                    rows.add(r)
                    r += rng.randrange(1, 7)
        if shape.single:
            rows = {r for r in rows if r + 3 not in block}
        elif not disjoint_selectors:
            # every pair of gate selectors shares a row: halo2's compress_selectors then leaves each its own fixed column (the
            # layout zk_keygen builds keys for; selectors that never meet would be COMBINED into one column — ZK_ELAYOUT)
            rows.add(common_row)
        sel_rows.append(sorted(rows))
        col = fixed[shape.fx_sel[j]]
        for r in rows:
            col[r] = 1
    is_output = [[False] * n for _ in range(shape.n_adv)]
    for j in range(A):
        for r in sel_rows[j]:
            is_output[j][r + 3] = True

    # ---- small-class cells (must hold table values)
    small = [[False] * n for _ in range(shape.n_adv)]
    if shape.single:
        ql = fixed[shape.fx_qlookup]
        for r in block:
            ql[r] = 1
            small[0][r] = True
    else:
        for l in range(shape.n_lookup_cols):
            for r in range(usable):
                small[A + l][r] = True  # every row of a lookup column is looked up; only the block is non-zero

    # ---- copy cycles over all permutation columns.  perm col: fixed f -> f, advice j -> F + j
    def perm_col(kind, i):
        return i if kind == "fixed" else F + i

    used = set()
    force_small = set()
    cycles = []  # [(members sorted by (row, col) with the source first), value class]
    free_cells = [(j, r) for j in range(shape.n_adv) for r in range(usable) if not is_output[j][r] and j != zero_col
                  and (j < A or r in block)]
    n_cycles = max(6, usable // 6)
    for ci in range(n_cycles):
        size = rng.choice((2, 2, 3, 5, 8, 13)) if ci % 7 else rng.randrange(25, 41)
        members = []
        for _ in range(size * 3):
            c = free_cells[rng.randrange(len(free_cells))]
            if c not in used and c not in members:
                members.append(c)
            if len(members) == size:
                break
        if len(members) < 2:
            continue
        members.sort(key=lambda c: (c[1], c[0]))
        need_small = any(small[j][r] for j, r in members)
        kind = rng.random()
        src = None
        if kind < 0.3:  # sourced by a constants cell
            src = ("fixed", rng.randrange(F), rng.randrange(usable))
            if need_small:
                fixed[src[1]][src[2]] = rng.randrange(T)
        elif kind < 0.5 and not need_small:  # sourced by a gate output above every member
            outs = [(j, r) for j in range(A) if j != zero_col for r in range(3, members[0][1]) if is_output[j][r] and (j, r) not in used]
            if outs:
                o = outs[rng.randrange(len(outs))]
                src = ("advice", o[0], o[1])
                used.add(o)
        for c in members:
            used.add(c)
        if src is None and need_small:
            force_small.add(members[0])  # the cycle's own first cell is its source: it must hold a table value
        cycles.append((src, members, need_small))
    # a zero cycle through the all-zero column, zero constants and zero cells elsewhere
    zero_cycle = None
    if zero_col is not None:
        zc = [(zero_col, r) for r in rng.sample(range(usable), 6)]
        others = [c for c in free_cells if c not in used and not (shape.single and c[1] in block)][:3]
        fixed[0][1] = 0
        zero_cycle = (("fixed", 0, 1), zc + others, False)
        for c in others:
            used.add(c)

    # ---- witness, in row order: a cell is a gate output, a cycle member (source's value) or free
    src_of = {}
    for src, members, need_small in cycles:
        head = members[0] if src is None else None
        for c in members:
            if c != head:
                src_of[c] = src if src is not None else ("advice", head[0], head[1])
    if zero_cycle:
        for c in zero_cycle[1]:
            src_of[c] = zero_cycle[0]
    gate_at = [set(rows) for rows in sel_rows]
    for r in range(usable):
        for j in range(shape.n_adv):
            if j == zero_col:
                continue
            if j < A and r >= 3 and (r - 3) in gate_at[j]:
                advice[j][r] = (advice[j][r - 3] + advice[j][r - 2] * advice[j][r - 1]) % R
            elif (j, r) in src_of:
                k, i, rr = src_of[(j, r)]
                advice[j][r] = fixed[i][rr] if k == "fixed" else advice[i][rr]
            elif small[j][r] or (j, r) in force_small:
                advice[j][r] = rng.randrange(T) if (r in block or (j, r) in force_small) else 0
            else:
                advice[j][r] = rng.choice((rng.randrange(T), rng.randrange(R), rng.randrange(1 << 88), 0, 1, R - 1))
    copies = []
    for src, members, _ in cycles + ([zero_cycle] if zero_cycle else []):
        chain = ([(perm_col("fixed", src[1]), src[2])] if src and src[0] == "fixed" else
                 [(perm_col("advice", src[1]), src[2])] if src else []) + [(perm_col("advice", j), r) for j, r in members]
        rng.shuffle(chain)  # the order copies are declared in is not the row order
        for a, b in zip(chain, chain[1:]):
            copies.append((a, b))
    return fixed, copies, advice


def check(shape, fixed, copies, advice):
    """The assignment satisfies gates, lookups and copies (plain Python, independent of the provers)."""
    F, A, usable = shape.num_fixed, shape.n_gate, shape.usable_rows
    T = 1 << shape.lookup_bits
    for j in range(A):
        for r in range(usable):
            if fixed[shape.fx_sel[j]][r]:
                assert (advice[j][r] + advice[j][r + 1] * advice[j][r + 2] - advice[j][r + 3]) % R == 0, (j, r)
    cell = lambda c: fixed[c[0]][c[1]] if c[0] < F else advice[c[0] - F][c[1]]
    for a, b in copies:
        assert cell(a) == cell(b), (a, b)
    for r in range(usable):
        if shape.single:
            assert fixed[shape.fx_qlookup][r] * advice[0][r] % R < T
        else:
            for l in range(shape.n_lookup_cols):
                assert advice[A + l][r] < T
