#!/usr/bin/env python3
"""Generates sparksmithwaterman_amd/csrc/swmi_cells_gen.inc: the hand-scheduled gfx950 instruction
stream that updates the R cells a lane owns in one anti-diagonal step of the fill kernel.

Per cell (ACGT variant) 8 VALU instructions:
    v_dot8_i32_i4 a, q, rb, diag         a = NW + s(ref,read): rb is the reference symbol one-hot (1 << 4*symbol), q the
                                         row's 8 x int4 score profile, the accumulator operand the diagonal neighbour
    v_cmp_ge_i32  sI, up, left           insertion beats deletion?   (v_cmp_gt for the DistributedSW order)
    v_max_i32     t, up, left
    v_add_u32     t, gap, t              t = max(N, W) + gap
    v_cmp_ge_i32  sA, a, t               alignment beats both?       (v_cmp_gt for the DistributedSW order)
    v_addc_co_u32 acc, vcc, acc, acc, sI   acc = 2*acc + bit  (compare mask used directly as carry-in)
    v_max3_i32    hout, a, t, 0          (not in place: the previous column's H stays readable for one more step,
                                          which lets the tied-maximum check be deferred instead of stalling on it)
    v_addc_co_u32 acc, vcc, acc, acc, sA
gfx950 needs 2 wait states between a VALU that writes an SGPR pair and a VALU that reads it; the rows
are interleaved so every compare has at least two independent instructions before its v_addc, and the
script checks that distance (inserting s_nop only where a variant cannot avoid it).

The generic variant (any byte alphabet, or scores outside int8) replaces the profile lookup by
v_cmp_eq_u32 + v_cndmask_b32.
"""
import os
import sys

INTERLEAVE = os.environ.get("SWMI_GEN_INTERLEAVE", "1") != "0"
HAZARD = 2   # wait states: VALU writes SGPR -> VALU reads that SGPR (gfx940/gfx950)
DOT_HAZARD = 3   # wait states: v_dot* writes a VGPR -> a different VALU opcode reads it (gfx90a+; LLVM's
                 # DotWriteDifferentVALURead).  Inline asm is opaque to the compiler's hazard recognizer.


class Ins:
    def __init__(self, text, wr=(), rd=(), dot=()):
        self.text, self.wr, self.rd, self.dot = text, tuple(wr), tuple(rd), tuple(dot)


def schedule(R, acgt, strict, dirs=True):
    ge = "v_cmp_gt_i32_e64" if strict else "v_cmp_ge_i32_e64"
    ins = []
    # phase 1: substitution scores and the diagonal candidates (all from the previous column's values)
    def nw(k):
        return "%[diag]" if k == 0 else f"%[i{k-1}]"
    if acgt:
        # rb is the reference symbol ONE-HOT (1 << 4*symbol): the dot product with the row's 8 x int4 score profile
        # picks s(ref, read), and the accumulator operand adds the diagonal neighbour in the same instruction
        for k in range(R):
            ins.append(Ins(f"v_dot8_i32_i4 %[a{k}], %[q{k}], %[rb], {nw(k)}", dot=[f"a{k}"]))
    else:
        for k in range(R):
            ins.append(Ins(f"v_cmp_eq_u32_e64 %[m{k}], %[rb], %[q{k}]", wr=[f"m{k}"]))
        for k in range(R):
            ins.append(Ins(f"v_cndmask_b32_e64 %[a{k}], %[vmis], %[vmat], %[m{k}]", rd=[f"m{k}"]))
    if not acgt:
        for k in range(R):
            ins.append(Ins(f"v_add_u32_e32 %[a{k}], %[a{k}], {nw(k)}"))
    # phase 2: the dependent chain down the lane's rows
    if not dirs:
        # score-only variant (checkpoint/recompute mode): 5 VALU per cell, no direction bits.  The row chain
        # max -> add -> max3 is strictly dependent; the independent profile lookups of the rows below are woven
        # into it (phase 1 above only emitted row 0's) so that dependent instructions are not back to back.
        chain = []
        for k in range(R):
            up = "%[up]" if k == 0 else f"%[o{k-1}]"
            chain.append([Ins(f"v_max_i32_e32 %[t], {up}, %[i{k}]"),
                          Ins(f"v_add_u32_e32 %[t], %[gap], %[t]"),
                          Ins(f"v_max3_i32 %[o{k}], %[a{k}], %[t], 0")])
        if INTERLEAVE and acgt:
            # ins currently holds: bfe x R, add x R.  Rebuild: row 0 lookup first, then weave.
            look = [[Ins(f"v_dot8_i32_i4 %[a{k}], %[q{k}], %[rb], {nw(k)}", dot=[f"a{k}"])] for k in range(R)]
            ins = list(look[0])
            pending = [x for k in range(1, R) for x in look[k]]
            for k in range(R):
                for c in chain[k]:
                    ins.append(c)
                    if pending:
                        # row k+1's a must be complete before its max3: keep at least its two lookups ahead
                        ins.append(pending.pop(0))
            ins.extend(pending)
        else:
            for k in range(R):
                ins.extend(chain[k])
    else:
        for k in range(R):
            up = "%[up]" if k == 0 else f"%[o{k-1}]"
            ins.append(Ins(f"{ge} %[sI], {up}, %[i{k}]", wr=["sI"]))
            if k > 0:
                ins.append(Ins(f"v_addc_co_u32_e64 %[acc{k-1}], vcc, %[acc{k-1}], %[acc{k-1}], %[sA]", rd=["sA"]))
            ins.append(Ins(f"v_max_i32_e32 %[t], {up}, %[i{k}]"))
            ins.append(Ins(f"v_add_u32_e32 %[t], %[gap], %[t]"))
            ins.append(Ins(f"{ge} %[sA], %[a{k}], %[t]", wr=["sA"]))
            ins.append(Ins(f"v_addc_co_u32_e64 %[acc{k}], vcc, %[acc{k}], %[acc{k}], %[sI]", rd=["sI"]))
            ins.append(Ins(f"v_max3_i32 %[o{k}], %[a{k}], %[t], 0"))
        ins.append(Ins(f"v_addc_co_u32_e64 %[acc{R-1}], vcc, %[acc{R-1}], %[acc{R-1}], %[sA]", rd=["sA"]))
    # hazard pass
    out = []
    last_wr = {}
    last_dot = {}
    for i in ins:
        need = 0
        for r in i.rd:
            if r in last_wr:
                gap = len(out) - last_wr[r] - 1 + sum(1 for _ in ())
                # count issued instructions (s_nop N counts N+1 states)
                states = 0
                for o in out[last_wr[r] + 1:]:
                    states += o.states if hasattr(o, "states") else 1
                need = max(need, HAZARD - states)
        for r, at in last_dot.items():
            if not i.dot and ("%[" + r + "]") in i.text.split(",", 1)[-1]:      # a source operand of a non-dot VALU
                states = 0
                for o in out[at + 1:]:
                    states += o.states if hasattr(o, "states") else 1
                need = max(need, DOT_HAZARD - states)
        if need > 0:
            nop = Ins(f"s_nop {need - 1}")
            nop.states = need
            out.append(nop)
        out.append(i)
        for r in i.wr:
            last_wr[r] = len(out) - 1
        for r in i.dot:
            last_dot[r] = len(out) - 1
    return out


def emit(R, acgt, strict, dirs=True):
    body = schedule(R, acgt, strict, dirs)
    n_valu = sum(1 for i in body if not i.text.startswith("s_nop"))
    n_nop = len(body) - n_valu
    lines = []
    lines.append(f"// R={R} {'ACGT' if acgt else 'GENERIC'} {'STRICT' if strict else 'SERIAL'} {'DIRS' if dirs else 'SCORE-ONLY'}: "
                 f"{n_valu} VALU ({n_valu / R:.1f}/cell), {n_nop} s_nop")
    lines.append("template <> struct CellsAsm<%d, %s, %s, %s> {" % (R, "true" if acgt else "false", "true" if strict else "false", "true" if dirs else "false"))
    lines.append("    static __device__ __forceinline__ void step(const int (&hin)[%d], int (&hout)[%d], uint32_t (&acc)[%d], const int (&q)[%d]," % (R, R, R, R))
    lines.append("                                                int rb, int diag, int up, int gap, int vmat, int vmis) {")
    lines.append("        int " + ", ".join(f"a{k}" for k in range(R)) + ", t;")
    sg = (["sI", "sA"] if dirs else []) + ([] if acgt else [f"m{k}" for k in range(R)])
    if sg:
        lines.append("        unsigned long long " + ", ".join(sg) + ";")
    lines.append("        asm volatile(")
    for i in body:
        lines.append(f'            "{i.text}\\n\\t"')
    outs = [f'[o{k}] "=&v"(hout[{k}])' for k in range(R)] + ([f'[acc{k}] "+v"(acc[{k}])' for k in range(R)] if dirs else [])
    outs += [f'[a{k}] "=&v"(a{k})' for k in range(R)] + ['[t] "=&v"(t)'] + [f'[{s}] "=&s"({s})' for s in sg]
    ins_ = [f'[i{k}] "v"(hin[{k}])' for k in range(R)] + [f'[q{k}] "v"(q[{k}])' for k in range(R)] + ['[rb] "v"(rb)', '[diag] "v"(diag)', '[up] "v"(up)', '[gap] "s"(gap)']
    if not acgt:
        ins_ += ['[vmat] "v"(vmat)', '[vmis] "v"(vmis)']
    lines.append("            : " + ", ".join(outs))
    lines.append("            : " + ", ".join(ins_))
    lines.append('            : "vcc");' if dirs else '            : );')
    if not dirs:
        lines.append("        (void)acc;")
    if acgt:
        lines.append("        (void)vmat; (void)vmis;")
    lines.append("    }")
    lines.append("};")
    return "\n".join(lines)


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "sparksmithwaterman_amd", "csrc", "swmi_cells_gen.inc")
    parts = ["// GENERATED by tools/gen_cells.py -- do not edit; re-run the script instead.",
             "// Direction bits pushed per cell: first bI (insertion >= deletion), then bA (alignment >= both):",
             "// the 2-bit code is (bI << 1) | bA; traceback decodes A if bit0, else I if bit1, else D.",
             "template <int R, bool ACGT, bool STRICT, bool DIRS> struct CellsAsm;", ""]
    for R in (1, 2, 3, 4):
        for acgt in (True, False):
            for strict in (False, True):
                parts.append(emit(R, acgt, strict, True))
                parts.append("")
            parts.append(emit(R, acgt, False, False))     # scores do not depend on the tie order
            parts.append("")
    text = "\n".join(parts)
    if "--check" in sys.argv:                       # tests/test_abi.py: the committed file is what this script generates
        same = os.path.exists(path) and open(path).read() == text
        print("up to date" if same else "STALE: re-run tools/gen_cells.py", path)
        return 0 if same else 1
    with open(path, "w") as f:
        f.write(text)
    print("wrote", path)


if __name__ == "__main__":
    sys.exit(main())
