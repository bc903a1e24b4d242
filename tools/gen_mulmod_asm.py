#!/usr/bin/env python3
"""Generates ecfft_amd/csrc/secp256k1_mul_gfx950.inc: the hand-scheduled gfx950 instruction streams of
t*x (+c) mod p for secp256k1 (p = 2^256 - 2^32 - 977), as C++ inline-asm statements.

Scheme (DESIGN.md "Field arithmetic on the CDNA4 VALU"):
  product   64 x v_mad_u64_u32: partial product t_i*x_j (column k=i+j) accumulates into the 64-bit VGPR
            pair that STARTS at word k — E[k/2] for even k, O[(k-1)/2] for odd k — so nothing is ever
            shifted or realigned (64-bit VGPR operands must be even-aligned on gfx950).  Each mad's
            carry-out (vcc) is counted by one v_addc_co_u32 into a per-pair overflow word.
            One asm statement per row i (8 mads) keeps hipcc's post-asm s_nop padding to 8.
  combine   W = E + (O << 32) + overflow words: two v_addc carry chains.
  reduce    V = lo + (hi << 32) + hi*977 (8 mads, no overflow), second fold of the 34-bit top,
            conditional subtract of p via "+ (2^32+977), keep if it carried".
"""
import os

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ecfft_amd", "csrc", "secp256k1_mul_gfx950.inc")


def pair_name(k):
    return f"E{k // 2}" if k % 2 == 0 else f"O{(k - 1) // 2}"


GROUP = int(os.environ.get("ECFFT_MUL_GROUP", "0"))   # >0: carries go to distinct SGPR pairs, mads issued in groups of GROUP before their addc's


def gen_product(has_c):
    """returns list of C++ lines; defines E0..E7, O0..O6 (uint64_t) and X<k> overflow words (uint32_t)"""
    lines = []
    lines.append("        uint64_t E0, E1, E2, E3, E4, E5, E6, E7, O0, O1, O2, O3, O4, O5, O6;")
    nprod = {k: min(k, 14 - k) + 1 for k in range(15)}
    seen = {k: 0 for k in range(15)}          # products already added to pair k
    ex_live = {k: False for k in range(15)}
    ex_decl = []
    for i in range(8):
        outs, ins, body = [], [], []

        def op_out(name, cons):
            outs.append((name, cons)); return len(outs) - 1

        # build operand list: outputs first (pairs, ex), then inputs
        pair_idx, ex_idx = {}, {}
        for j in range(8):
            k = i + j
            first = seen[k] == 0
            pair_idx[k] = op_out(pair_name(k), "=&v" if first else "+v")
        # overflow words
        for j in range(8):
            k = i + j
            first = seen[k] == 0
            can_overflow = (not first) or (has_c and k % 2 == 0 and k < 8)
            if can_overflow:
                if not ex_live[k]:
                    ex_idx[k] = (op_out(f"X{k}", "=&v"), True)
                    ex_decl.append(f"X{k}")
                else:
                    ex_idx[k] = (op_out(f"X{k}", "+v"), False)
        nout = len(outs)
        in_names = [f"t.l[{i}]"] + [f"x.l[{j}]" for j in range(8)]
        c_idx = {}
        if has_c and i == 0:
            for q in range(4):
                c_idx[2 * q] = nout + len(in_names); in_names.append(f"c{q}")
        t_op = nout
        if GROUP:
            # carry-outs to distinct SGPR pairs (asm temporaries), so that mads of different accumulators do not
            # serialise on VCC; the addc's of a group follow the group's mads
            ncar = 8
            car0 = len(outs)
            for q in range(ncar):
                outs.append((f"cy{q}", "=&s"))
            shift = ncar
            nout2 = len(outs)
            def inop(idx):   # input operand numbers move up by the extra outputs
                return idx + shift
            t_op2 = inop(t_op)
            pend = []
            for j in range(8):
                k = i + j
                first = seen[k] == 0
                P = pair_idx[k]
                xj = inop(nout + 1 + j)
                cy = car0 + j
                if first and k in c_idx:
                    body.append(f"v_mad_u64_u32 %{P}, %{cy}, %{t_op2}, %{xj}, %{inop(c_idx[k])}")
                elif first:
                    body.append(f"v_mad_u64_u32 %{P}, %{cy}, %{t_op2}, %{xj}, 0")
                else:
                    body.append(f"v_mad_u64_u32 %{P}, %{cy}, %{t_op2}, %{xj}, %{P}")
                if k in ex_idx:
                    xi, fresh = ex_idx[k]
                    pend.append((xi, fresh, cy))
                    ex_live[k] = True
                seen[k] += 1
                if (j + 1) % GROUP == 0 or j == 7:
                    for xi, fresh, cyq in pend:
                        if fresh:
                            body.append(f"v_addc_co_u32_e64 %{xi}, %{cyq}, 0, 0, %{cyq}")
                        else:
                            body.append(f"v_addc_co_u32_e64 %{xi}, %{cyq}, 0, %{xi}, %{cyq}")
                    pend = []
        else:
          for j in range(8):
            k = i + j
            first = seen[k] == 0
            P = pair_idx[k]
            xj = nout + 1 + j
            if first and k in c_idx:
                body.append(f"v_mad_u64_u32 %{P}, vcc, %{t_op}, %{xj}, %{c_idx[k]}")
            elif first:
                body.append(f"v_mad_u64_u32 %{P}, vcc, %{t_op}, %{xj}, 0")
            else:
                body.append(f"v_mad_u64_u32 %{P}, vcc, %{t_op}, %{xj}, %{P}")
            if k in ex_idx:
                xi, fresh = ex_idx[k]
                if fresh:
                    body.append(f"v_addc_co_u32_e64 %{xi}, vcc, 0, 0, vcc")
                else:
                    body.append(f"v_addc_co_u32_e32 %{xi}, vcc, 0, %{xi}, vcc")
                ex_live[k] = True
            seen[k] += 1
        asm = "\\n\\t".join(body)
        outs_s = ", ".join(f'"{c}"({n})' for n, c in outs)
        ins_s = ", ".join(f'"v"({n})' for n in in_names)
        lines.append(f'        asm("{asm}"\n            : {outs_s}\n            : {ins_s}\n            : "vcc");')
    assert all(seen[k] == nprod[k] for k in range(15))
    decl = "        uint32_t " + ", ".join(ex_decl) + ";"
    if GROUP:
        decl += "\n        uint64_t cy0, cy1, cy2, cy3, cy4, cy5, cy6, cy7;"
    if has_c:
        cdecl = ["        const uint64_t c%d = (uint64_t)c.l[%d] | ((uint64_t)c.l[%d] << 32);" % (q, 2 * q, 2 * q + 1) for q in range(4)]
    else:
        cdecl = []
    return cdecl + [lines[0], decl] + lines[1:], ex_live


def gen_combine(ex_live):
    """W = E + (O<<32) + X.  word k of E: e[k]; of O<<32: o[k] (k=1..14); overflow word of pair k sits at word k+2."""
    L = []
    e = [f"(uint32_t)(E{k // 2}{' >> 32' if k % 2 else ''})" for k in range(16)]
    o = {k: f"(uint32_t)(O{(k - 1) // 2}{' >> 32' if (k - 1) % 2 else ''})" for k in range(1, 15)}
    # chain 1: s[k] = e[k] + o[k] + cy, fresh outputs
    L.append("        uint32_t s0 = " + e[0] + ";")
    L.append("        uint32_t " + ", ".join(f"s{k}" for k in range(1, 16)) + ";")
    body = []
    for k in range(1, 16):
        so = k - 1
        ei = 15 + (k - 1)
        if k == 1:
            body.append(f"v_add_co_u32_e32 %{so}, vcc, %{ei}, %{15 + 15 + 0}")
        elif k <= 14:
            body.append(f"v_addc_co_u32_e32 %{so}, vcc, %{ei}, %{15 + 15 + (k - 1)}, vcc")
        else:
            body.append(f"v_addc_co_u32_e32 %{so}, vcc, 0, %{ei}, vcc")
    outs = ", ".join(f'"=&v"(s{k})' for k in range(1, 16))
    ins = ", ".join([f'"v"({e[k]})' for k in range(1, 16)] + [f'"v"({o[k]})' for k in range(1, 15)])
    L.append('        asm("' + "\\n\\t".join(body) + f'"\n            : {outs}\n            : {ins}\n            : "vcc");')
    # chain 2: s[k] += X[k-2] for live overflow words; carries propagate through every word up to 15
    xs = {k + 2: f"X{k}" for k in range(15) if ex_live[k] and k + 2 <= 15}
    lo = min(xs)
    body, outs_l, ins_l = [], [], []
    for k in range(lo, 16):
        outs_l.append(f'"+v"(s{k})')
    nout = len(outs_l)
    in_pos = {}
    for k in sorted(xs):
        in_pos[k] = nout + len(ins_l); ins_l.append(f'"v"({xs[k]})')
    for k in range(lo, 16):
        so = k - lo
        src = f"%{in_pos[k]}" if k in in_pos else "0"
        if k == lo:
            body.append(f"v_add_co_u32_e32 %{so}, vcc, {src}, %{so}")
        else:
            body.append(f"v_addc_co_u32_e32 %{so}, vcc, {src}, %{so}, vcc")
    L.append('        asm("' + "\\n\\t".join(body) + f'"\n            : {", ".join(outs_l)}\n            : {", ".join(ins_l)}\n            : "vcc");')
    return L


def gen_reduce():
    """s0..s15 -> canonical r (8 words).  One asm statement; temporaries are early-clobber outputs."""
    L = []
    L.append("        uint64_t ue0, ue1, ue2, ue3, uo0, uo1, uo2, uo3, m;")
    L.append("        uint32_t v0, v1, v2, v3, v4, v5, v6, v7, v8, v9;")
    # statement R1: the 8 mads hi*977 (no overflow)
    body = []
    outs = [("ue0", "=&v"), ("uo0", "=&v"), ("ue1", "=&v"), ("uo1", "=&v"), ("ue2", "=&v"), ("uo2", "=&v"), ("ue3", "=&v"), ("uo3", "=&v")]
    ins = [f"s{k}" for k in range(8, 16)] + ["k977"]
    for j in range(8):
        body.append(f"v_mad_u64_u32 %{j}, vcc, %{8 + j}, %16, 0")
    L.append("        const uint32_t k977 = 977u;")
    L.append('        asm("' + "\\n\\t".join(body) + '"\n            : ' + ", ".join(f'"{c}"({n})' for n, c in outs) +
             "\n            : " + ", ".join(f'"v"({n})' for n in ins[:8]) + ', "s"(k977)\n            : "vcc");')
    # statement R2: V = lo + (hi<<32)  [chain A], += U_E [chain B], += U_O<<32 [chain C]
    ue = [f"(uint32_t)(ue{k // 2}{' >> 32' if k % 2 else ''})" for k in range(8)]          # word k
    uo = {k: f"(uint32_t)(uo{(k - 1) // 2}{' >> 32' if (k - 1) % 2 else ''})" for k in range(1, 9)}  # word k
    outs = [f'"=&v"(v{k})' for k in range(10)]
    ins = []
    pos = {}

    def inp(expr):
        if expr not in pos:
            pos[expr] = 10 + len(ins); ins.append(f'"v"({expr})')
        return f"%{pos[expr]}"
    body = []
    # chain A
    body.append(f"v_add_co_u32_e32 %1, vcc, {inp('s1')}, {inp('s8')}")
    for k in range(2, 8):
        body.append(f"v_addc_co_u32_e32 %{k}, vcc, {inp(f's{k}')}, {inp(f's{7 + k}')}, vcc")
    body.append(f"v_addc_co_u32_e32 %8, vcc, 0, {inp('s15')}, vcc")
    body.append("v_addc_co_u32_e64 %9, vcc, 0, 0, vcc")
    # chain B (word 0 starts from s0)
    body.append(f"v_add_co_u32_e32 %0, vcc, {inp('s0')}, {inp(ue[0])}")
    for k in range(1, 8):
        body.append(f"v_addc_co_u32_e32 %{k}, vcc, {inp(ue[k])}, %{k}, vcc")
    body.append("v_addc_co_u32_e32 %8, vcc, 0, %8, vcc")
    body.append("v_addc_co_u32_e32 %9, vcc, 0, %9, vcc")
    # chain C
    body.append(f"v_add_co_u32_e32 %1, vcc, {inp(uo[1])}, %1")
    for k in range(2, 9):
        body.append(f"v_addc_co_u32_e32 %{k}, vcc, {inp(uo[k])}, %{k}, vcc")
    body.append("v_addc_co_u32_e32 %9, vcc, 0, %9, vcc")
    L.append('        asm("' + "\\n\\t".join(body) + '"\n            : ' + ", ".join(outs) + "\n            : " + ", ".join(ins) + '\n            : "vcc");')
    # statement R3: second fold + canonicalise
    #   m = v8*977 ; m.hi += v9*977 (v9 <= 3) ; R = v[0..7] + m + ({v9:v8} << 32) ; overflow o
    #   Rc = R + (2^32+977) ; take Rc if (o | carry)
    L.append("        uint32_t r0, r1, r2, r3, r4, r5, r6, r7, q0, q1, q2, q3, q4, q5, q6, q7, t9;")
    L.append("        uint64_t ovf, ovf2;")
    body = []
    # operands: outputs r0..r7 (0-7), q0..q7 (8-15), m (16), t9 (17), ovf (18, sgpr pair), ovf2 (19); inputs v0..v9 (20-29), k977 (30)
    body.append("v_mad_u64_u32 %16, vcc, %28, %30, 0")                 # m = v8*977
    body.append("v_mul_u32_u24_e32 %17, %30, %29")                     # t9 = 977*v9  (v9 <= 3)   [src0 may be sgpr]
    # chain D: R = v + m  (word0: m.lo, word1: m.hi + t9 folded below)
    # first add t9 into m.hi without carry loss: m.hi < 977+1, t9 < 2^12 -> plain add
    # we cannot address halves of %16, so add (t9 << 32) through a second chain on word 1 instead.
    L_m_lo = "(uint32_t)m"; L_m_hi = "(uint32_t)(m >> 32)"
    # --> split: statement R3a computes m and t9; statement R3b consumes halves
    L.append('        asm("v_mad_u64_u32 %0, vcc, %2, %4, 0\\n\\tv_mul_u32_u24_e32 %1, %4, %3"\n            : "=&v"(m), "=&v"(t9)\n            : "v"(v8), "v"(v9), "s"(k977)\n            : "vcc");')
    body = []
    # R3b operands: outputs r0..r7 (0-7), q0..q7 (8-15), ovf (16 "=&s"), ovf2 (17 "=&s");
    #               inputs v0..v7 (18-25), v8 (26), v9 (27), m.lo (28), m.hi (29), t9 (30)
    body.append("v_add_co_u32_e32 %0, vcc, %28, %18")                   # r0 = v0 + m.lo
    body.append("v_addc_co_u32_e32 %1, vcc, %29, %19, vcc")             # r1 = v1 + m.hi + cy
    for k in range(2, 8):
        body.append(f"v_addc_co_u32_e32 %{k}, vcc, 0, %{18 + k}, vcc")  # r_k = v_k + cy
    body.append("s_mov_b64 %16, vcc")                                   # ovf  = carry out of chain D
    # chain E: + ((t9 + v8) at word 1, v9 at word 2): t9 + v8 may carry -> do two adds
    body.append("v_add_co_u32_e32 %1, vcc, %26, %1")                    # r1 += v8
    body.append("v_addc_co_u32_e32 %2, vcc, %27, %2, vcc")              # r2 += v9 + cy
    for k in range(3, 8):
        body.append(f"v_addc_co_u32_e32 %{k}, vcc, 0, %{k}, vcc")
    body.append("s_or_b64 %16, %16, vcc")
    body.append("v_add_co_u32_e32 %1, vcc, %30, %1")                    # r1 += t9
    for k in range(2, 8):
        body.append(f"v_addc_co_u32_e32 %{k}, vcc, 0, %{k}, vcc")
    body.append("s_or_b64 %16, %16, vcc")
    # q = r + (2^32 + 977)
    body.append("v_add_co_u32_e32 %8, vcc, 0x3d1, %0")
    body.append("v_addc_co_u32_e32 %9, vcc, 1, %1, vcc")
    for k in range(2, 8):
        body.append(f"v_addc_co_u32_e32 %{8 + k}, vcc, 0, %{k}, vcc")
    body.append("s_or_b64 %17, %16, vcc")                               # take q if overflowed before or r >= p
    for k in range(8):
        body.append(f"v_cndmask_b32_e64 %{k}, %{k}, %{8 + k}, %17")
    outs = [f'"=&v"(r{k})' for k in range(8)] + [f'"=&v"(q{k})' for k in range(8)] + ['"=&s"(ovf)', '"=&s"(ovf2)']
    ins = [f'"v"(v{k})' for k in range(8)] + ['"v"(v8)', '"v"(v9)', '"v"((uint32_t)m)', '"v"((uint32_t)(m >> 32))', '"v"(t9)']
    # s_or_b64 writes SCC: it must be declared, hipcc keeps loop compares live across asm statements
    L.append('        asm("' + "\\n\\t".join(body) + '"\n            : ' + ", ".join(outs) + "\n            : " + ", ".join(ins) + '\n            : "vcc", "scc");')
    L.append("        elem r; r.l[0] = r0; r.l[1] = r1; r.l[2] = r2; r.l[3] = r3; r.l[4] = r4; r.l[5] = r5; r.l[6] = r6; r.l[7] = r7;")
    L.append("        return r;")
    return L


def main():
    out = ["// GENERATED by tools/gen_mulmod_asm.py — do not edit.  Included inside struct ecfft::Secp256k1 (device only).", ""]
    for has_c in (False, True):
        name = "mul_add_gfx950" if has_c else "mul_gfx950"
        sig = "const elem& t, const elem& x, const elem& c" if has_c else "const elem& t, const elem& x"
        out.append(f"    __device__ static inline elem {name}({sig}) {{")
        prod, ex_live = gen_product(has_c)
        out += prod
        out += gen_combine(ex_live)
        out += gen_reduce()
        out.append("    }")
        out.append("")
    with open(OUT, "w") as f:
        f.write("\n".join(out))
    print("wrote", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
