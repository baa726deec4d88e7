#!/usr/bin/env python3
"""Generates sparksmithwaterman_amd/csrc/swmi_step_gen.inc: the anti-diagonal steps of the mode-1 score sweep
(sw_sweep_winmax_kernel, fast symbols, single strip) as a hand-scheduled gfx950 instruction stream, FOUR steps per
asm statement (hipcc pads every boundary between two asm statements with an s_nop of its own) -- per step the lane's
R cells, both neighbour exchanges, the reference feed and the window maximum.

State of a lane (swmi_kernels.hip, SweepFast): H of its R rows ping-pongs between two register sets (hin = the
previous step's, hout = the one before, overwritten here); hp[k] = max(H[k] + gap, 0) is kept beside it, in place.
With gap <= 0 the recurrence of SmithWaterman.java:223-249
    H = max(0, W + gap, N + gap, NW + s)        becomes        H = max3(NW + s, hp(N), hp(W))
because hp >= 0 already carries the clamp: 3 VALU per cell instead of 4,
    v_dot8_i32_i4  a, q, rb, diag        a = NW + s(ref,read)   (one-hot symbol . row profile, + diagonal)
    v_max3_i32     H, a, hp_up, hp_left
    v_sub_u32      hp, H, |gap| clamp    unsigned saturating: max(H - |gap|, 0)
Row 0 takes its neighbours from lane l-1 through the DPP network INSIDE the arithmetic (no separate v_mov_dpp):
    v_dot8_i32_i4  s0, q0, rb, 0
    v_add_u32_dpp  a0, hout[R-1], s0  wave_shr:1 bound_ctrl   NW = lane l-1's bottom row two steps ago (lane 0: 0)
    v_max_i32_dpp  x,  hp[R-1],  hp0  wave_shr:1 bound_ctrl   max(hp(N), hp(W))          (lane 0: N = 0)
    v_max_i32      H0, a0, x
The one-hot reference symbol of the NEXT step is prepared here (v_lshlrev_b32_sdwa feed for lane 0 + one v_mov_b32_dpp
shift), so a step never starts with a dependent pair.  The per-lane window maximum takes v_max3 on 2 values at a time:
an even step leaves its last row for the odd one when R is odd.

Hazards the script enforces by construction (it simulates the previous step's tail in front of each stream and
pads with s_nop where an instruction order cannot avoid it):
  * v_dot* result read by another VALU opcode: 3 wait states;
  * a VGPR written by a VALU and read through DPP (or overwritten by a DPP mov): 2 wait states.
"""
import os
import sys

# Rows 1 .. R-1 accumulate NW + s IN PLACE with the 4-byte VOP2 form v_dot8c_i32_i4 (dst += dot): the accumulator is the
# previous step's H of the row above, which nothing reads again (W comes from hp, the next step overwrites the whole set, and
# the neighbour lane only ever reads row R-1).  Same instruction count, 4 bytes less per cell: with several wavefronts per SIMD
# the 8-byte encodings issue at half the rate of the 4-byte ones (tools/ubench_occ.hip), so big batches gain; one wavefront per
# SIMD is indifferent.  SWMI_GEN_DOT8C=0 generates the three-address form.
DOT8C = os.environ.get("SWMI_GEN_DOT8C", "1") != "0"
DOT_WAIT = 3
DPP_WAIT = 2
SGPR_WAIT = 2     # gfx940/gfx950: a VALU writes an SGPR, a VALU reads it


class Ins:
    def __init__(self, text, wr=(), rd=(), dot=(), dpp_rd=(), dpp_wr=(), sg_wr=(), sg_rd=()):
        self.text = text
        self.wr, self.rd, self.dot, self.dpp_rd, self.dpp_wr = tuple(wr), tuple(rd), tuple(dot), tuple(dpp_rd), tuple(dpp_wr)
        self.sg_wr, self.sg_rd = tuple(sg_wr), tuple(sg_rd)      # SGPR pairs written by a VALU (v_cmp) / read by a VALU (v_addc)
        self.states = 1


def stream(R, odd, feed_byte, tail=False):
    """instruction list of one step; register names are asm operand names.  Even steps read h / write g and consume
    rbx / prepare rby; odd steps the other way round.  tail: the step of a block in which some lanes have run past the last
    column -- they compute values nobody reads, but their window maximum must not see them: it is updated under a lane
    mask (v_cmp once per step, v_cndmask per update)."""
    H, G = ("g", "h") if odd else ("h", "g")
    RB, RBN = ("rby", "rbx") if odd else ("rbx", "rby")
    WF = "wf1" if feed_byte == 0 else "wf0"          # the last step of a group of four is fed from the next dword
    hin = lambda k: f"%[{H}{k}]"
    hout = lambda k: f"%[{G}{k}]"
    hp = lambda k: f"%[p{k}]"
    dpp = "wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
    ins = []
    if tail:
        ins.append(Ins("v_cmp_lt_u32_e32 vcc, %[c], %[n]", wr=["vcc"], rd=["c", "n"]))       # this lane's column is inside the reference
    ins.append(Ins(f"v_dot8_i32_i4 %[s0], %[q0], %[{RB}], 0", wr=["s0"], dot=["s0"], rd=["q0", RB]))
    an = lambda k: f"{H}{k-1}" if DOT8C else f"a{k}"             # where a_k = NW + s of row k lives
    for k in range(1, R):
        if DOT8C:
            ins.append(Ins(f"v_dot8c_i32_i4_e32 {hin(k-1)}, %[q{k}], %[{RB}]", wr=[an(k)], dot=[an(k)], rd=[f"q{k}", RB, f"{H}{k-1}"]))
        else:
            ins.append(Ins(f"v_dot8_i32_i4 %[a{k}], %[q{k}], %[{RB}], {hin(k-1)}", wr=[f"a{k}"], dot=[f"a{k}"], rd=[f"q{k}", RB, f"{H}{k-1}"]))
    ins.append(Ins(f"v_lshlrev_b32_sdwa %[{RBN}], %[{WF}], %[one] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_{feed_byte} src1_sel:DWORD",
                   wr=[RBN], rd=[WF, "one"]))
    ins.append(Ins(f"v_max_i32_dpp %[x], {hp(R-1)}, {hp(0)} {dpp}", wr=["x"], rd=["p0"], dpp_rd=[f"p{R-1}"]))
    ins.append(Ins(f"v_add_u32_dpp %[a0], {hout(R-1)}, %[s0] {dpp}", wr=["a0"], rd=["s0"], dpp_rd=[f"{G}{R-1}"]))
    ins.append(Ins(f"v_mov_b32_dpp %[{RBN}], %[{RB}] wave_shr:1 row_mask:0xf bank_mask:0xf", dpp_rd=[RB], dpp_wr=[RBN], wr=[RBN]))
    ins.append(Ins(f"v_max_i32_e32 {hout(0)}, %[a0], %[x]", wr=[f"{G}0"], rd=["a0", "x"]))
    ins.append(Ins(f"v_sub_u32_e64 {hp(0)}, {hout(0)}, %[gm] clamp", wr=["p0"], rd=[f"{G}0"]))
    for k in range(1, R):
        ins.append(Ins(f"v_max3_i32 {hout(k)}, %[{an(k)}], {hp(k-1)}, {hp(k)}", wr=[f"{G}{k}"], rd=[an(k), f"p{k-1}", f"p{k}"]))
        ins.append(Ins(f"v_sub_u32_e64 {hp(k)}, {hout(k)}, %[gm] clamp", wr=[f"p{k}"], rd=[f"{G}{k}"]))
    # window maximum: even steps consume 2*floor(R/2) of their values, odd steps the leftover row of the even step + their own
    vals = [hout(k) for k in range(R)]
    if tail:
        # no deferral: a lane may be inside the reference at an even step and outside at the odd one, which would drop the row
        # it left over -- every step takes all of its own values (the last one twice when R is odd)
        if R % 2:
            vals.append(vals[-1])
    elif not odd:
        vals = vals[:2 * (R // 2)]
    elif R % 2:
        vals = [hin(R - 1)] + vals
    for j in range(0, len(vals), 2):
        if tail:
            ins.append(Ins(f"v_max3_i32 %[x], %[lm], {vals[j]}, {vals[j+1]}", wr=["x"], rd=["lm"]))
            ins.append(Ins("v_cndmask_b32_e32 %[lm], %[lm], %[x], vcc", wr=["lm"], rd=["lm", "x", "vcc"]))
        else:
            ins.append(Ins(f"v_max3_i32 %[lm], %[lm], {vals[j]}, {vals[j+1]}", wr=["lm"], rd=["lm"]))
    return ins


def pad_hazards(prev_tail, body):
    """inserts s_nop into `body` so that every hazard holds, given the instructions that ran just before it"""
    out = list(prev_tail)
    base = len(out)
    for i in body:
        need = 0
        def states_since(pos):
            return sum(o.states for o in out[pos + 1:])
        for pos in range(len(out) - 1, -1, -1):
            o = out[pos]
            if states_since(pos) >= max(DOT_WAIT, DPP_WAIT):
                break
            if not i.dot:
                for r in o.dot:
                    if r in i.rd or r in i.dpp_rd:
                        need = max(need, DOT_WAIT - states_since(pos))
            for r in o.wr:
                if r in i.dpp_rd or r in i.dpp_wr:
                    need = max(need, DPP_WAIT - states_since(pos))
            for r in o.sg_wr:
                if r in i.sg_rd:
                    need = max(need, SGPR_WAIT - states_since(pos))
        if need > 0:
            nop = Ins(f"s_nop {need - 1}")
            nop.states = need
            out.append(nop)
        out.append(i)
    return out[base:]


def emit(R):
    """four consecutive steps (phases 0..3 of a 16-step block) as ONE asm statement: the compiler pads every boundary
    between two asm statements with an s_nop of its own"""
    group = []
    for ph in range(4):
        group += stream(R, ph & 1, (ph + 1) & 3)
    prev = stream(R, 1, 0)                             # the step before the group is phase 3 of the previous one
    body = pad_hazards(prev[-4:], group)
    n_nop = sum(1 for i in body if i.text.startswith("s_nop"))
    n_valu = len(body) - n_nop
    L = []
    L.append(f"// R={R}: four steps, {n_valu} VALU, {n_nop} s_nop  ({n_valu / 4:.2f} VALU per step, {n_valu / 4 / R:.2f} per cell)")
    L.append(f"template <> struct SweepStep4Asm<{R}> {{")
    L.append(f"    static __device__ __forceinline__ void run(int (&h)[{R}], int (&g)[{R}], int (&hp)[{R}], const int (&q)[{R}],")
    L.append(f"                                               int &rbx, int &rby, const uint32_t wf0, const uint32_t wf1,")
    L.append(f"                                               const int one, const uint32_t gm, int &lm) {{")
    L.append("        int s0, a0, x" + "".join(f", a{k}" for k in range(1, R)) + ";")
    L.append("        asm volatile(")
    for i in body:
        L.append(f'            "{i.text}\\n\\t"')
    outs = [f'[h{k}] "+v"(h[{k}])' for k in range(R)] + [f'[g{k}] "+v"(g[{k}])' for k in range(R)]
    outs += [f'[p{k}] "+v"(hp[{k}])' for k in range(R)]
    outs += ['[rbx] "+v"(rbx)', '[rby] "+v"(rby)', '[lm] "+v"(lm)', '[s0] "=&v"(s0)', '[a0] "=&v"(a0)', '[x] "=&v"(x)']
    outs += [f'[a{k}] "=&v"(a{k})' for k in range(1, R)]
    ins_ = [f'[q{k}] "v"(q[{k}])' for k in range(R)]
    ins_ += ['[wf0] "v"(wf0)', '[wf1] "v"(wf1)', '[one] "v"(one)', '[gm] "s"(gm)']
    L.append("            : " + ", ".join(outs))
    L.append("            : " + ", ".join(ins_))
    L.append("            : );")
    L.append("    }")
    L.append("};")
    return "\n".join(L), n_valu, n_nop


def dir_stream(R, odd, feed_byte, strict):
    """one step of the re-sweep WITH direction bits (the traceback's window replay): the 3-VALU cell plus two compares whose
    lane masks feed v_addc as carry-in (acc = 2 * acc + bit), first `insertion beats deletion`, then `alignment beats both`:
        serial ('>=' chain, SmithWaterman.java:223-249):  bI = hp(N) >= hp(W),  bA = (H == a)   [a >= max(hp, hp) <=> a is the max]
        strict ('>' chain, DistributedSW.java:305-330):   bI = hp(N) >  hp(W),  bA = a > max(hp(N), hp(W))
    On the clamped values hp = max(H + gap, 0) the compares only differ from the unclamped ones where both candidates are <= 0,
    i.e. where the cell's direction is the alignment or its H is 0 and never read."""
    H, G = ("g", "h") if odd else ("h", "g")
    RB, RBN = ("rby", "rbx") if odd else ("rbx", "rby")
    WF = "wf1" if feed_byte == 0 else "wf0"
    hin = lambda k: f"%[{H}{k}]"
    hout = lambda k: f"%[{G}{k}]"
    hp = lambda k: f"%[p{k}]"
    dpp = "wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
    cmpI = "v_cmp_gt_u32_e64" if strict else "v_cmp_ge_u32_e64"
    ins = []
    ins.append(Ins(f"v_dot8_i32_i4 %[s0], %[q0], %[{RB}], 0", wr=["s0"], dot=["s0"], rd=["q0", RB]))
    for k in range(1, R):
        if DOT8C:
            ins.append(Ins(f"v_dot8c_i32_i4_e32 {hin(k-1)}, %[q{k}], %[{RB}]", wr=[f"{H}{k-1}"], dot=[f"{H}{k-1}"], rd=[f"q{k}", RB, f"{H}{k-1}"]))
        else:
            ins.append(Ins(f"v_dot8_i32_i4 %[a{k}], %[q{k}], %[{RB}], {hin(k-1)}", wr=[f"a{k}"], dot=[f"a{k}"], rd=[f"q{k}", RB, f"{H}{k-1}"]))
    ins.append(Ins(f"v_lshlrev_b32_sdwa %[{RBN}], %[{WF}], %[one] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_{feed_byte} src1_sel:DWORD",
                   wr=[RBN], rd=[WF, "one"]))
    ins.append(Ins(f"v_mov_b32_dpp %[x], {hp(R-1)} {dpp}", wr=["x"], dpp_rd=[f"p{R-1}"]))                      # hp(N) of row 0
    ins.append(Ins(f"v_add_u32_dpp %[a0], {hout(R-1)}, %[s0] {dpp}", wr=["a0"], rd=["s0"], dpp_rd=[f"{G}{R-1}"]))
    ins.append(Ins(f"v_mov_b32_dpp %[{RBN}], %[{RB}] wave_shr:1 row_mask:0xf bank_mask:0xf", dpp_rd=[RB], dpp_wr=[RBN], wr=[RBN]))
    # the second addc of a row waits two states for its compare: it is issued inside the NEXT row (and the last row's
    # inside the next step, after its dot products -- emit_dir moves `trailing` there)
    pending = None
    for k in range(R):
        up = "%[x]" if k == 0 else hp(k - 1)
        upn = "x" if k == 0 else f"p{k-1}"
        ak = "a0" if k == 0 else (f"{H}{k-1}" if DOT8C else f"a{k}")
        sI, sA = (("sI0", "sA0") if k % 2 == 0 else ("sI1", "sA1"))
        ins.append(Ins(f"{cmpI} %[{sI}], {up}, {hp(k)}", rd=[upn, f"p{k}"], sg_wr=[sI]))
        if strict:
            ins.append(Ins(f"v_max_u32_e32 %[t], {up}, {hp(k)}", wr=["t"], rd=[upn, f"p{k}"]))
        ins.append(Ins(f"v_max3_i32 {hout(k)}, %[{ak}], {up}, {hp(k)}", wr=[f"{G}{k}"], rd=[ak, upn, f"p{k}"]))
        if pending:
            ins.append(pending)
        ins.append(Ins(f"v_sub_u32_e64 {hp(k)}, {hout(k)}, %[gm] clamp", wr=[f"p{k}"], rd=[f"{G}{k}"]))
        if strict:
            ins.append(Ins(f"v_cmp_gt_i32_e64 %[{sA}], %[{ak}], %[t]", rd=[ak, "t"], sg_wr=[sA]))
        else:
            ins.append(Ins(f"v_cmp_eq_u32_e64 %[{sA}], {hout(k)}, %[{ak}]", rd=[f"{G}{k}", ak], sg_wr=[sA]))
        ins.append(Ins(f"v_addc_co_u32_e64 %[c{k}], vcc, %[c{k}], %[c{k}], %[{sI}]", wr=[f"c{k}"], rd=[f"c{k}"], sg_rd=[sI]))
        pending = Ins(f"v_addc_co_u32_e64 %[c{k}], vcc, %[c{k}], %[c{k}], %[{sA}]", wr=[f"c{k}"], rd=[f"c{k}"], sg_rd=[sA])
    return ins, pending


def emit_dir(R, strict):
    group = []
    trailing = None
    for ph in range(4):
        body, last = dir_stream(R, ph & 1, (ph + 1) & 3, strict)
        if trailing:
            body.insert(min(R, 2), trailing)           # after the step's first dot products
        group += body
        trailing = last
    group.append(trailing)
    prev, plast = dir_stream(R, 1, 0, strict)
    body = pad_hazards((prev + [plast])[-4:], group)
    n_nop = sum(1 for i in body if i.text.startswith("s_nop"))
    n_valu = len(body) - n_nop
    L = []
    L.append(f"// R={R} {'STRICT' if strict else 'SERIAL'}: four steps with direction bits, {n_valu} VALU, {n_nop} s_nop  ({n_valu / 4:.2f} per step, {n_valu / 4 / R:.2f} per cell)")
    L.append(f"template <> struct DirStep4Asm<{R}, {'true' if strict else 'false'}> {{")
    L.append(f"    static __device__ __forceinline__ void run(int (&h)[{R}], int (&g)[{R}], int (&hp)[{R}], uint32_t (&acc)[{R}], const int (&q)[{R}],")
    L.append(f"                                               int &rbx, int &rby, const uint32_t wf0, const uint32_t wf1,")
    L.append(f"                                               const int one, const uint32_t gm) {{")
    L.append("        int s0, a0, x" + (", t" if strict else "") + "".join(f", a{k}" for k in range(1, R)) + ";")
    L.append("        unsigned long long sI0, sA0, sI1, sA1;")
    L.append("        asm volatile(")
    for i in body:
        L.append(f'            "{i.text}\\n\\t"')
    outs = [f'[h{k}] "+v"(h[{k}])' for k in range(R)] + [f'[g{k}] "+v"(g[{k}])' for k in range(R)]
    outs += [f'[p{k}] "+v"(hp[{k}])' for k in range(R)] + [f'[c{k}] "+v"(acc[{k}])' for k in range(R)]
    outs += ['[rbx] "+v"(rbx)', '[rby] "+v"(rby)', '[s0] "=&v"(s0)', '[a0] "=&v"(a0)', '[x] "=&v"(x)']
    if strict:
        outs += ['[t] "=&v"(t)']
    outs += [f'[a{k}] "=&v"(a{k})' for k in range(1, R)]
    outs += ['[sI0] "=&s"(sI0)', '[sA0] "=&s"(sA0)', '[sI1] "=&s"(sI1)', '[sA1] "=&s"(sA1)']
    ins_ = [f'[q{k}] "v"(q[{k}])' for k in range(R)]
    ins_ += ['[wf0] "v"(wf0)', '[wf1] "v"(wf1)', '[one] "v"(one)', '[gm] "s"(gm)']
    L.append("            : " + ", ".join(outs))
    L.append("            : " + ", ".join(ins_))
    L.append('            : "vcc");')
    L.append("    }")
    L.append("};")
    return "\n".join(L), n_valu, n_nop


def emit_tail(R, ph):
    """one masked step (phase ph of a block) as its own asm statement: the few blocks at the end of a sweep"""
    odd = ph & 1
    body = pad_hazards(stream(R, not odd, ph, True)[-4:], stream(R, odd, (ph + 1) & 3, True))
    n_nop = sum(1 for i in body if i.text.startswith("s_nop"))
    L = []
    L.append(f"// R={R} phase {ph}: one masked step, {len(body) - n_nop} VALU, {n_nop} s_nop")
    L.append(f"template <> struct SweepStepTailAsm<{R}, {ph}> {{")
    L.append(f"    static __device__ __forceinline__ void run(int (&h)[{R}], int (&g)[{R}], int (&hp)[{R}], const int (&q)[{R}],")
    L.append(f"                                               int &rbx, int &rby, const uint32_t wf0, const uint32_t wf1,")
    L.append(f"                                               const int one, const uint32_t gm, int &lm, const uint32_t c, const uint32_t n) {{")
    L.append("        int s0, a0, x" + "".join(f", a{k}" for k in range(1, R)) + ";")
    L.append("        asm volatile(")
    for i in body:
        L.append(f'            "{i.text}\\n\\t"')
    outs = [f'[h{k}] "+v"(h[{k}])' for k in range(R)] + [f'[g{k}] "+v"(g[{k}])' for k in range(R)]
    outs += [f'[p{k}] "+v"(hp[{k}])' for k in range(R)]
    outs += ['[rbx] "+v"(rbx)', '[rby] "+v"(rby)', '[lm] "+v"(lm)', '[s0] "=&v"(s0)', '[a0] "=&v"(a0)', '[x] "=&v"(x)']
    outs += [f'[a{k}] "=&v"(a{k})' for k in range(1, R)]
    ins_ = [f'[q{k}] "v"(q[{k}])' for k in range(R)]
    ins_ += ['[wf0] "v"(wf0)', '[wf1] "v"(wf1)', '[one] "v"(one)', '[gm] "s"(gm)', '[c] "v"(c)', '[n] "v"(n)']
    L.append("            : " + ", ".join(outs))
    L.append("            : " + ", ".join(ins_))
    L.append('            : "vcc");')
    L.append("    }")
    L.append("};")
    return "\n".join(L)


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "sparksmithwaterman_amd", "csrc", "swmi_step_gen.inc")
    parts = ["// GENERATED by tools/gen_step.py -- do not edit; re-run the script instead.",
             "// Four anti-diagonal steps of the mode-1 score sweep per specialisation (see the script's docstring).",
             "template <int R> struct SweepStep4Asm;", ""]
    for R in (1, 2, 3, 4):
        text, nv, nn = emit(R)
        parts.append(text)
        parts.append("")
        print(f"R={R}: {(nv + nn) / 4:.2f} instructions per step ({(nv + nn) / 4 / R:.2f} per cell), {nn} s_nop per 4 steps")
    parts += ["// The same step for the blocks at the end of a sweep, where lanes run past the last column: one step per statement,",
              "// the window maximum updated under a lane mask (see the script's docstring).",
              "template <int R, int PH> struct SweepStepTailAsm;", ""]
    for R in (1, 2, 3, 4):
        for ph in range(4):
            parts.append(emit_tail(R, ph))
            parts.append("")
    parts += ["// The re-sweep with direction bits (window replay of the traceback), four steps per statement.",
              "template <int R, bool STRICT> struct DirStep4Asm;", ""]
    for R in (1, 2, 3, 4):
        for strict in (False, True):
            text, nv, nn = emit_dir(R, strict)
            parts.append(text)
            parts.append("")
            print(f"R={R} {'strict' if strict else 'serial'} with direction bits: {(nv + nn) / 4:.2f} instructions per step, {nn} s_nop per 4 steps")
    text = "\n".join(parts)
    if "--check" in sys.argv:                       # tests/test_abi.py: the committed file is what this script generates
        same = os.path.exists(path) and open(path).read() == text
        print("up to date" if same else "STALE: re-run tools/gen_step.py", path)
        return 0 if same else 1
    with open(path, "w") as f:
        f.write(text)
    print("wrote", path)


if __name__ == "__main__":
    sys.exit(main())
