"""Static VALU instruction mix of every kernel of the library, from `hipcc -S` (runs in the build container, no GPU):
    python tools/isa_mix.py [tag]            -> profiles/<tag>_isa_mix.csv
Per kernel: vector instructions by issue class and the mix-weighted issue cycles per wave-instruction per SIMD, with the class
rates MEASURED on MI355X at four waves per SIMD (tools/ubench_isa.hip, profiles/r2_ubench_isa.txt):
    v_mad_u64_u32 4.72   v_mul_lo/hi_u32 4.3   carry chain 4.39   64-bit shifts / adds / moves 4.56   v_fma_f64 4.62
    VOP3-encoded 32-bit instructions (v_add3, v_alignbit, v_bfe, v_lshl_add, v_and_or, ...) 4.3   plain VOP1 / VOP2 2.5 (3.96 between multiply-adds)
bench.py's roofline.valu_issue multiplies the per-kernel instruction counts of the PMC pass (SQ_INSTS_VALU) by these weights to
price a proof's instruction stream in SIMD issue cycles.  The mix is the whole kernel's static text: the hot loops of these
kernels are fully unrolled (a mixed addition is 2 000 straight-line instructions), so their text IS what executes; rarely taken
paths (the general addition inside the accumulation's fallback, error exits) are a few percent of the text.
Also prints, for the bucket accumulation, the instruction counts of the mixed-addition basic block (the review's "non-mad
instructions per addition")."""
import collections
import csv
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "webauthn-halo2_amd", "csrc")
FILES = ["msm.hip", "ntt.hip", "quotient.hip", "poly.hip", "prover_kernels.hip", "engine.hip", "serde.hip"]
CXXFILT = "c++filt"  # (binutils; llvm-cxxfilt is not in the image)

# issue cycles per wave-instruction per SIMD at the nominal 2.4 GHz, four waves per SIMD, every SIMD busy (profiles/r6_ubench_isa.txt;
# round 2's file has the first nine).  Only runs of plain 32-bit VOP1 / VOP2 instructions reach the double rate; everything
# encoded as VOP3 (three operands: v_add3, v_alignbit, v_bfe, v_lshl_add, v_and_or, v_mad_u32_u24, 64-bit moves and shifts)
# issues at the multiplier's rate — and a plain instruction BETWEEN two multiply-adds costs ~ 4.0, not 2.5 ("4 mad + 4 v_and":
# 4.34 per instruction): `simple` below is therefore the optimistic price, SIMPLE_MIXED the one measured in a mixed stream.
CLASS_CYCLES = {"mad64": 4.72, "mul32": 4.3, "carry": 4.39, "wide64": 4.56, "f64": 4.62, "vop3": 4.3, "simple": 2.5}
SIMPLE_MIXED = 3.96


def classify(op):
    if op.startswith("v_mad_u64_u32") or op.startswith("v_mad_i64_i32"):
        return "mad64"
    if op.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mul_lo_i32", "v_mul_u32_u24", "v_mul_hi_u32_u24")):
        return "mul32"
    if op.startswith(("v_addc_co", "v_subb_co", "v_subbrev_co", "v_add_co", "v_sub_co", "v_subrev_co")):
        return "carry"
    if op.startswith(("v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64", "v_lshl_add_u64", "v_mov_b64", "v_add_u64", "v_sub_u64")):
        return "wide64"
    if op.endswith("_f64") or "_f64_" in op:
        return "f64"
    if op.endswith("_e32") or op in ("v_nop",):
        return "simple"
    return "vop3"  # _e64 encodings and the VOP3-only instructions (v_add3_u32, v_alignbit_b32, v_bfe_u32, v_lshl_add_u32, v_perm_b32, ...)


def kernels_of(asm_path):
    """{mangled name: [instruction mnemonics]} for every .amdhsa_kernel of an assembly file, with the basic-block labels kept."""
    text = open(asm_path).read().splitlines()
    names = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", "\n".join(text), flags=re.M))
    out, cur = {}, None
    for line in text:
        m = re.match(r"^(\S+):\s*(;.*)?$", line)
        if m and m.group(1) in names:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        s = line.strip()
        if s.startswith(".Lfunc_end") or s.startswith(".section"):
            cur = None
            continue
        if not s or s.startswith((";", ".")) and not s.startswith(".LBB"):
            continue
        if s.startswith(".LBB"):
            cur.append(("label", s.split(":")[0]))
            continue
        cur.append(("op", s.split()[0]))
    return out


def demangle(names):
    p = subprocess.run([CXXFILT], input="\n".join(names), stdout=subprocess.PIPE, text=True)
    return dict(zip(names, (l.split("(")[0].strip() for l in p.stdout.splitlines())))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
    tmp = "/tmp/isa_mix"
    os.makedirs(tmp, exist_ok=True)
    rows = []
    for f in FILES:
        s = os.path.join(tmp, f.replace(".hip", ".s"))
        src = os.path.join(CSRC, f)
        if not os.path.exists(s) or os.path.getmtime(s) < max(os.path.getmtime(os.path.join(CSRC, x)) for x in os.listdir(CSRC)):
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-S", "--cuda-device-only", "-o", s, src])
        ks = kernels_of(s)
        dm = demangle(list(ks))
        for mangled, seq in ks.items():
            cnt = collections.Counter()
            nops = 0
            for kind, op in seq:
                if kind != "op":
                    continue
                if op.startswith("v_"):
                    cnt[classify(op)] += 1
                elif op == "s_nop":
                    nops += 1
            total = sum(cnt.values())
            if not total:
                continue
            cyc = sum(cnt[c] * CLASS_CYCLES[c] for c in cnt)
            cyc_mixed = cyc + cnt["simple"] * (SIMPLE_MIXED - CLASS_CYCLES["simple"])
            rows.append([dm[mangled], f, total, cnt["mad64"], cnt["mul32"], cnt["carry"], cnt["wide64"], cnt["f64"], cnt["vop3"], cnt["simple"], nops,
                         "%.3f" % (cyc / total), "%.3f" % (cyc_mixed / total)])
            if "msm_wacc_fast_kernel" in mangled:
                # the mixed addition = the largest basic block
                blocks, cur = [], []
                for kind, op in seq:
                    if kind == "label":
                        blocks.append(cur)
                        cur = []
                    else:
                        cur.append(op)
                blocks.append(cur)
                big = max(blocks, key=len)
                c2 = collections.Counter(op for op in big if op.startswith("v_"))
                mads = c2["v_mad_u64_u32"]
                print("msm_wacc_fast_kernel, mixed-addition block: %d VALU instructions = %d v_mad_u64_u32 + %d others; %d s_nop"
                      % (sum(c2.values()), mads, sum(c2.values()) - mads, sum(1 for op in big if op == "s_nop")))
                for op, n in c2.most_common():
                    if op != "v_mad_u64_u32":
                        print("    %-22s %d" % (op, n))
    rows.sort(key=lambda r: -r[2])
    out = os.path.join(ROOT, "profiles", "%s_isa_mix.csv" % tag)
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "file", "valu_static", "mad64", "mul32", "carry", "wide64", "f64", "vop3", "simple", "s_nop",
                    "issue_cycles_per_valu_instruction", "issue_cycles_per_valu_instruction_mixed_stream"])
        w.writerows(rows)
    print("wrote", out, "(%d kernels)" % len(rows))


if __name__ == "__main__":
    main()
