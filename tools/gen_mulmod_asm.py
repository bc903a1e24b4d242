#!/usr/bin/env python3
"""Generates ecfft_amd/csrc/secp256k1_mul_gfx950.inc: the hand-scheduled gfx950 instruction streams of
t*x (+c) mod p for secp256k1 (p = 2^256 - 2^32 - 977), as C++ inline-asm statements.

Scheme (DESIGN.md "Field arithmetic on the CDNA4 VALU"):
  (three asm statements per multiply keep hipcc's post-asm s_nop padding to 3)
  product   64 x v_mad_u64_u32: partial product t_i*x_j (column k=i+j) accumulates into the 64-bit VGPR
            pair that STARTS at word k — E[k/2] for even k, O[(k-1)/2] for odd k — so nothing is ever
            shifted or realigned (64-bit VGPR operands must be even-aligned on gfx950).  Each mad's
            carry-out (vcc) is counted by one v_addc_co_u32 into a per-pair overflow word.
  combine   W = E + (O << 32) + overflow words: two v_addc carry chains.
  reduce    V = lo + (hi << 32) + hi*977 (8 mads, no overflow), second fold of the 34-bit top,
            conditional subtract of p via "+ (2^32+977), keep if it carried".
"""
import os

OUT = os.environ.get("ECFFT_MUL_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ecfft_amd", "csrc", "secp256k1_mul_gfx950.inc")


def pair_name(k):
    return f"E{k // 2}" if k % 2 == 0 else f"O{(k - 1) // 2}"


GROUP = int(os.environ.get("ECFFT_MUL_GROUP", "0"))   # >0: carries go to distinct SGPR pairs, mads issued in groups of GROUP before their addc's


SPLIT = int(os.environ.get("ECFFT_MUL_SPLIT", "2"))   # asm statements for the 64-mad product (register pressure vs s_nop padding)


def gen_product(has_c):
    """SPLIT asm statements (8/SPLIT rows of 8 mads each); defines E0..E7, O0..O6 (uint64_t) and X<k> overflow words.
    One statement needs every operand live at once (early-clobber outputs cannot share registers with inputs), which
    costs ~20 VGPRs in the callers; two statements are the measured optimum on MI355X."""
    nprod = {k: min(k, 14 - k) + 1 for k in range(15)}
    seen = {k: 0 for k in range(15)}
    ex_live = {k: False for k in range(15)}
    need_ex = {k: (nprod[k] + (1 if (has_c and k % 2 == 0 and k < 8) else 0)) >= 2 for k in range(15)}
    lines = []
    if has_c:
        lines += ["        const uint64_t c%d = (uint64_t)c.l[%d] | ((uint64_t)c.l[%d] << 32);" % (q, 2 * q, 2 * q + 1) for q in range(4)]
    lines.append("        uint64_t E0, E1, E2, E3, E4, E5, E6, E7, O0, O1, O2, O3, O4, O5, O6;")
    lines.append("        uint32_t " + ", ".join(f"X{k}" for k in range(15) if need_ex[k]) + ";")
    rows_per = 8 // SPLIT
    for part in range(SPLIT):
        rows = range(part * rows_per, (part + 1) * rows_per)
        ks = sorted({i + j for i in rows for j in range(8)})
        outs, pair_idx, ex_idx = [], {}, {}
        for k in ks:
            pair_idx[k] = len(outs); outs.append((pair_name(k), "=&v" if seen[k] == 0 else "+v"))
        for k in ks:
            # does this part add to pair k when it could overflow?
            will = False; cnt = seen[k]
            for i in rows:
                j = k - i
                if 0 <= j < 8:
                    if cnt > 0 or (has_c and k % 2 == 0 and k < 8):
                        will = True
                    cnt += 1
            if will:
                ex_idx[k] = len(outs); outs.append((f"X{k}", "+v" if ex_live[k] else "=&v"))
        nout = len(outs)
        in_names, tpos, xpos, c_idx = [], {}, {}, {}
        for i in rows:
            tpos[i] = nout + len(in_names); in_names.append(f"t.l[{i}]")
        for j in range(8):
            xpos[j] = nout + len(in_names); in_names.append(f"x.l[{j}]")
        if has_c and part == 0:
            for q in range(4):
                c_idx[2 * q] = nout + len(in_names); in_names.append(f"c{q}")
        body = []
        for i in rows:
            for j in range(8):
                k = i + j
                first = seen[k] == 0
                P = pair_idx[k]
                if first and k in c_idx:
                    body.append(f"v_mad_u64_u32 %{P}, vcc, %{tpos[i]}, %{xpos[j]}, %{c_idx[k]}")
                elif first:
                    body.append(f"v_mad_u64_u32 %{P}, vcc, %{tpos[i]}, %{xpos[j]}, 0")
                else:
                    body.append(f"v_mad_u64_u32 %{P}, vcc, %{tpos[i]}, %{xpos[j]}, %{P}")
                if (not first) or (k in c_idx):
                    xi = ex_idx[k]
                    if not ex_live[k]:
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
    return lines, ex_live


def gen_combine(ex_live):
    """statement 2: W = E + (O<<32) + X (two carry chains, fresh outputs s1..s15), then the eight mads hi*977.
    word k of E: e[k]; of O<<32: o[k] (k=1..14); overflow word of pair k sits at word k+2."""
    L = []
    e = [f"(uint32_t)(E{k // 2}{' >> 32' if k % 2 else ''})" for k in range(16)]
    o = {k: f"(uint32_t)(O{(k - 1) // 2}{' >> 32' if (k - 1) % 2 else ''})" for k in range(1, 15)}
    xs = {k + 2: f"X{k}" for k in range(15) if ex_live[k] and k + 2 <= 15}
    L.append("        uint32_t s0 = " + e[0] + ";")
    L.append("        uint32_t " + ", ".join(f"s{k}" for k in range(1, 16)) + ";")
    L.append("        uint64_t ue0, ue1, ue2, ue3, uo0, uo1, uo2, uo3;")
    L.append("        const uint32_t k977 = 977u;")
    outs = [f'"=&v"(s{k})' for k in range(1, 16)]                      # %0..%14 = s1..s15
    hi_names = ["ue0", "uo0", "ue1", "uo1", "ue2", "uo2", "ue3", "uo3"]   # hi word j (= s[8+j]) -> pair j
    outs += [f'"=&v"({n})' for n in hi_names]                          # %15..%22
    nout = len(outs)
    ins, pos = [], {}

    def inp(expr, cons="v"):
        if expr not in pos:
            pos[expr] = nout + len(ins); ins.append(f'"{cons}"({expr})')
        return f"%{pos[expr]}"
    body = []
    for k in range(1, 16):                                              # chain 1
        so = k - 1
        if k == 1:
            body.append(f"v_add_co_u32_e32 %{so}, vcc, {inp(e[k])}, {inp(o[k])}")
        elif k <= 14:
            body.append(f"v_addc_co_u32_e32 %{so}, vcc, {inp(e[k])}, {inp(o[k])}, vcc")
        else:
            body.append(f"v_addc_co_u32_e32 %{so}, vcc, 0, {inp(e[k])}, vcc")
    lo = min(xs)
    for k in range(lo, 16):                                             # chain 2
        so = k - 1
        src = inp(xs[k]) if k in xs else "0"
        if k == lo:
            body.append(f"v_add_co_u32_e32 %{so}, vcc, {src}, %{so}")
        else:
            body.append(f"v_addc_co_u32_e32 %{so}, vcc, {src}, %{so}, vcc")
    kk = inp("k977", "s")
    for j in range(8):                                                  # hi*977, no overflow possible
        body.append(f"v_mad_u64_u32 %{15 + j}, vcc, %{7 + j}, {kk}, 0")  # s[8+j] is output %(8+j-1)
    L.append('        asm("' + "\\n\\t".join(body) + f'"\n            : {", ".join(outs)}\n            : {", ".join(ins)}\n            : "vcc");')
    return L


def gen_reduce():
    """statement 3: V = lo + (hi<<32) + hi*977, second fold of the 34-bit top, canonicalise."""
    L = []
    L.append("        uint32_t v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, mlo, mhi, t9;")
    L.append("        uint32_t r0, r1, r2, r3, r4, r5, r6, r7, q0, q1, q2, q3, q4, q5, q6, q7;")
    L.append("        uint64_t ovf, ovf2;")
    ue = [f"(uint32_t)(ue{k // 2}{' >> 32' if k % 2 else ''})" for k in range(8)]          # word k
    uo = {k: f"(uint32_t)(uo{(k - 1) // 2}{' >> 32' if (k - 1) % 2 else ''})" for k in range(1, 9)}  # word k
    out_names = [f"v{k}" for k in range(10)] + ["mlo", "mhi", "t9"] + [f"r{k}" for k in range(8)] + [f"q{k}" for k in range(8)]
    outs = [f'"=&v"({n})' for n in out_names] + ['"=&s"(ovf)', '"=&s"(ovf2)']
    V = {k: k for k in range(10)}; MLO, MHI, T9 = 10, 11, 12
    Rr = {k: 13 + k for k in range(8)}; Q = {k: 21 + k for k in range(8)}; OVF, OVF2 = 29, 30
    nout = len(outs)
    ins, pos = [], {}

    def inp(expr, cons="v"):
        if expr not in pos:
            pos[expr] = nout + len(ins); ins.append(f'"{cons}"({expr})')
        return f"%{pos[expr]}"
    body = []
    # chain A: V = lo + (hi << 32)
    body.append(f"v_add_co_u32_e32 %{V[1]}, vcc, {inp('s1')}, {inp('s8')}")
    for k in range(2, 8):
        body.append(f"v_addc_co_u32_e32 %{V[k]}, vcc, {inp(f's{k}')}, {inp(f's{7 + k}')}, vcc")
    body.append(f"v_addc_co_u32_e32 %{V[8]}, vcc, 0, {inp('s15')}, vcc")
    body.append(f"v_addc_co_u32_e64 %{V[9]}, vcc, 0, 0, vcc")
    # chain B: += U_E
    body.append(f"v_add_co_u32_e32 %{V[0]}, vcc, {inp('s0')}, {inp(ue[0])}")
    for k in range(1, 8):
        body.append(f"v_addc_co_u32_e32 %{V[k]}, vcc, {inp(ue[k])}, %{V[k]}, vcc")
    body.append(f"v_addc_co_u32_e32 %{V[8]}, vcc, 0, %{V[8]}, vcc")
    body.append(f"v_addc_co_u32_e32 %{V[9]}, vcc, 0, %{V[9]}, vcc")
    # chain C: += U_O << 32
    body.append(f"v_add_co_u32_e32 %{V[1]}, vcc, {inp(uo[1])}, %{V[1]}")
    for k in range(2, 9):
        body.append(f"v_addc_co_u32_e32 %{V[k]}, vcc, {inp(uo[k])}, %{V[k]}, vcc")
    body.append(f"v_addc_co_u32_e32 %{V[9]}, vcc, 0, %{V[9]}, vcc")
    # second fold: top = {v9:v8} (< 2^34).  m = v8*977 (lo/hi), t9 = v9*977 (v9 <= 3)
    kk = inp("k977", "s")
    body.append(f"v_mul_lo_u32 %{MLO}, %{V[8]}, {kk}")
    body.append(f"v_mul_hi_u32 %{MHI}, %{V[8]}, {kk}")
    body.append(f"v_mul_u32_u24_e32 %{T9}, {kk}, %{V[9]}")
    # x1 = v8 + t9 (carry -> x2 = v9 + cy): reuse t9 and v9 registers
    body.append(f"v_add_co_u32_e32 %{T9}, vcc, %{V[8]}, %{T9}")
    body.append(f"v_addc_co_u32_e32 %{V[9]}, vcc, 0, %{V[9]}, vcc")
    # chain D: r = v[0..7] + {mhi:mlo}
    body.append(f"v_add_co_u32_e32 %{Rr[0]}, vcc, %{MLO}, %{V[0]}")
    body.append(f"v_addc_co_u32_e32 %{Rr[1]}, vcc, %{MHI}, %{V[1]}, vcc")
    for k in range(2, 8):
        body.append(f"v_addc_co_u32_e32 %{Rr[k]}, vcc, 0, %{V[k]}, vcc")
    body.append(f"s_mov_b64 %{OVF}, vcc")
    # chain E: r += {x2:x1} << 32
    body.append(f"v_add_co_u32_e32 %{Rr[1]}, vcc, %{T9}, %{Rr[1]}")
    body.append(f"v_addc_co_u32_e32 %{Rr[2]}, vcc, %{V[9]}, %{Rr[2]}, vcc")
    for k in range(3, 8):
        body.append(f"v_addc_co_u32_e32 %{Rr[k]}, vcc, 0, %{Rr[k]}, vcc")
    body.append(f"s_or_b64 %{OVF}, %{OVF}, vcc")
    # canonicalise: q = r + (2^32 + 977); take q if the value overflowed 2^256 before or r >= p.  r >= p needs r7 = 0xFFFFFFFF
    # (probability 2^-32 per lane) and an overflow is rarer still, so the 17-instruction select sits behind a wave-uniform
    # branch: 4 instructions on the fast path, identical results.
    body.append(f"v_cmp_eq_u32_e32 vcc, -1, %{Rr[7]}")
    body.append(f"s_or_b64 %{OVF2}, %{OVF}, vcc")
    body.append("s_cbranch_scc0 .Lecfft_canon_done_%=")
    body.append(f"v_add_co_u32_e32 %{Q[0]}, vcc, 0x3d1, %{Rr[0]}")
    body.append(f"v_addc_co_u32_e32 %{Q[1]}, vcc, 1, %{Rr[1]}, vcc")
    for k in range(2, 8):
        body.append(f"v_addc_co_u32_e32 %{Q[k]}, vcc, 0, %{Rr[k]}, vcc")
    body.append(f"s_or_b64 %{OVF2}, %{OVF}, vcc")
    for k in range(8):
        body.append(f"v_cndmask_b32_e64 %{Rr[k]}, %{Rr[k]}, %{Q[k]}, %{OVF2}")
    body.append(".Lecfft_canon_done_%=:")
    # s_or_b64 writes SCC: it must be declared, hipcc keeps loop compares live across asm statements
    L.append('        asm("' + "\\n\\t".join(body) + '"\n            : ' + ", ".join(outs) + "\n            : " + ", ".join(ins) + '\n            : "vcc", "scc");')
    L.append("        elem r; r.l[0] = r0; r.l[1] = r1; r.l[2] = r2; r.l[3] = r3; r.l[4] = r4; r.l[5] = r5; r.l[6] = r6; r.l[7] = r7;")
    L.append("        return r;")
    return L


# ----------------------------------------------------------------------------------------------------------------------
# Two-table variant: the table constant is stored as the pair (t, u = t * 2^128 mod p), so that
#   t*x = t*x_lo + u*x_hi  (x = x_lo + 2^128 x_hi)
# is a sum of two 256x128-bit products: still 64 mads, but the result has 12 words (+1 bit) instead of 16, the fold
# multiplies 4 high words instead of 8, and the value after the fold exceeds 2^256 only with probability ~2^-94, so the
# second fold moves into the (wave-uniform, essentially never taken) canonicalisation branch.
def gen_product2(has_c):
    npair = 11
    nprod = {k: 2 * min(k + 1, 4, 11 - k) for k in range(npair)}
    seen = {k: 0 for k in range(npair)}
    ex_live = {k: False for k in range(npair)}
    lines = []
    if has_c:
        lines += ["        const uint64_t c%d = (uint64_t)c.l[%d] | ((uint64_t)c.l[%d] << 32);" % (q, 2 * q, 2 * q + 1) for q in range(4)]
    lines.append("        uint64_t E0, E1, E2, E3, E4, E5, O0, O1, O2, O3, O4;")
    lines.append("        uint32_t " + ", ".join(f"X{k}" for k in range(npair)) + ";")
    for part in range(2):
        tn = "t" if part == 0 else "u"
        ks = list(range(npair))
        outs, pair_idx, ex_idx = [], {}, {}
        for k in ks:
            pair_idx[k] = len(outs); outs.append((pair_name(k), "=&v" if seen[k] == 0 else "+v"))
        for k in ks:
            will = (part == 1) or nprod[k] // 2 >= 2 or (has_c and k % 2 == 0 and k < 8)
            if will:
                ex_idx[k] = len(outs); outs.append((f"X{k}", "+v" if ex_live[k] else "=&v"))
        nout = len(outs)
        in_names, tpos, xpos, c_idx = [], {}, {}, {}
        for i in range(8):
            tpos[i] = nout + len(in_names); in_names.append(f"{tn}.l[{i}]")
        for j in range(4):
            xpos[j] = nout + len(in_names); in_names.append(f"x.l[{4 * part + j}]")
        if has_c and part == 0:
            for q in range(4):
                c_idx[2 * q] = nout + len(in_names); in_names.append(f"c{q}")
        body = []
        for i in range(8):
            for j in range(4):
                k = i + j
                first = seen[k] == 0
                P = pair_idx[k]
                if first and k in c_idx:
                    body.append(f"v_mad_u64_u32 %{P}, vcc, %{tpos[i]}, %{xpos[j]}, %{c_idx[k]}")
                elif first:
                    body.append(f"v_mad_u64_u32 %{P}, vcc, %{tpos[i]}, %{xpos[j]}, 0")
                else:
                    body.append(f"v_mad_u64_u32 %{P}, vcc, %{tpos[i]}, %{xpos[j]}, %{P}")
                if (not first) or (k in c_idx):
                    xi = ex_idx[k]
                    if not ex_live[k]:
                        body.append(f"v_addc_co_u32_e64 %{xi}, vcc, 0, 0, vcc")
                    else:
                        body.append(f"v_addc_co_u32_e32 %{xi}, vcc, 0, %{xi}, vcc")
                    ex_live[k] = True
                seen[k] += 1
        asm = "\\n\\t".join(body)
        outs_s = ", ".join(f'"{c}"({n})' for n, c in outs)
        ins_s = ", ".join(f'"v"({n})' for n in in_names)
        lines.append(f'        asm("{asm}"\n            : {outs_s}\n            : {ins_s}\n            : "vcc");')
    assert all(seen[k] == nprod[k] for k in range(npair)) and all(ex_live.values())
    return lines


def gen_combine2():
    """W = E + (O<<32) + X (13 words s0..s12, s12 <= 1), then the four mads hi*977 (words 8..11)."""
    L = []
    e = [f"(uint32_t)(E{k // 2}{' >> 32' if k % 2 else ''})" for k in range(12)]
    o = {k: f"(uint32_t)(O{(k - 1) // 2}{' >> 32' if (k - 1) % 2 else ''})" for k in range(1, 11)}
    xs = {k + 2: f"X{k}" for k in range(11)}
    L.append("        uint32_t s0 = " + e[0] + ";")
    L.append("        uint32_t " + ", ".join(f"s{k}" for k in range(1, 13)) + ";")
    L.append("        uint64_t ue0, ue1, uo0, uo1;")
    L.append("        const uint32_t k977 = 977u;")
    outs = [f'"=&v"(s{k})' for k in range(1, 13)]                      # %0..%11 = s1..s12
    hi_names = ["ue0", "uo0", "ue1", "uo1"]                            # s8*977 (words 0,1), s9*977 (1,2), s10*977 (2,3), s11*977 (3,4)
    outs += [f'"=&v"({n})' for n in hi_names]                          # %12..%15
    nout = len(outs)
    ins, pos = [], {}

    def inp(expr, cons="v"):
        if expr not in pos:
            pos[expr] = nout + len(ins); ins.append(f'"{cons}"({expr})')
        return f"%{pos[expr]}"
    body = []
    for k in range(1, 13):                                              # chain 1
        so = k - 1
        if k == 1:
            body.append(f"v_add_co_u32_e32 %{so}, vcc, {inp(e[k])}, {inp(o[k])}")
        elif k <= 10:
            body.append(f"v_addc_co_u32_e32 %{so}, vcc, {inp(e[k])}, {inp(o[k])}, vcc")
        elif k == 11:
            body.append(f"v_addc_co_u32_e32 %{so}, vcc, 0, {inp(e[k])}, vcc")
        else:
            body.append(f"v_addc_co_u32_e64 %{so}, vcc, 0, 0, vcc")
    for k in range(2, 13):                                              # chain 2
        so = k - 1
        if k == 2:
            body.append(f"v_add_co_u32_e32 %{so}, vcc, {inp(xs[k])}, %{so}")
        else:
            body.append(f"v_addc_co_u32_e32 %{so}, vcc, {inp(xs[k])}, %{so}, vcc")
    kk = inp("k977", "s")
    for j in range(4):
        body.append(f"v_mad_u64_u32 %{12 + j}, vcc, %{7 + j}, {kk}, 0")  # s[8+j] is output %(8+j-1)
    L.append('        asm("' + "\\n\\t".join(body) + f'"\n            : {", ".join(outs)}\n            : {", ".join(ins)}\n            : "vcc");')
    return L


def gen_reduce2():
    """V = lo + hi*(2^32 + 977), hi = s8..s12 (129 bits): H = hi*977 + (hi << 32) has 6 words, V = lo + H carries out of
    2^256 only if lo >= 2^256 - 2^163; that case and r >= p share the canonicalisation branch."""
    L = []
    L.append("        uint32_t h1, h2, h3, h4, h5, m4;")
    L.append("        uint32_t r0, r1, r2, r3, r4, r5, r6, r7, q0, q1, q2, q3, q4, q5, q6, q7;")
    L.append("        uint64_t ovf, ovf2;")
    ue = [f"(uint32_t)(ue{k // 2}{' >> 32' if k % 2 else ''})" for k in range(4)]           # word k, k = 0..3
    uo = {k: f"(uint32_t)(uo{(k - 1) // 2}{' >> 32' if (k - 1) % 2 else ''})" for k in range(1, 5)}  # word k, k = 1..4
    out_names = ["h1", "h2", "h3", "h4", "h5", "m4"] + [f"r{k}" for k in range(8)] + [f"q{k}" for k in range(8)]
    outs = [f'"=&v"({n})' for n in out_names] + ['"=&s"(ovf)', '"=&s"(ovf2)']
    H = {k: k - 1 for k in range(1, 6)}; M4 = 5
    Rr = {k: 6 + k for k in range(8)}; Q = {k: 14 + k for k in range(8)}; OVF, OVF2 = 22, 23
    nout = len(outs)
    ins, pos = [], {}

    def inp(expr, cons="v"):
        if expr not in pos:
            pos[expr] = nout + len(ins); ins.append(f'"{cons}"({expr})')
        return f"%{pos[expr]}"
    body = []
    kk = inp("k977", "s")
    # m4 = s12*977 + (word 4 of UO): both tiny
    body.append(f"v_mad_u32_u24 %{M4}, {inp('s12')}, {kk}, {inp(uo[4])}")
    # chain A: h[1..5] = (hi << 32) + (UO << 32) + m4 at word 4
    body.append(f"v_add_co_u32_e32 %{H[1]}, vcc, {inp('s8')}, {inp(uo[1])}")
    body.append(f"v_addc_co_u32_e32 %{H[2]}, vcc, {inp('s9')}, {inp(uo[2])}, vcc")
    body.append(f"v_addc_co_u32_e32 %{H[3]}, vcc, {inp('s10')}, {inp(uo[3])}, vcc")
    body.append(f"v_addc_co_u32_e32 %{H[4]}, vcc, {inp('s11')}, %{M4}, vcc")
    body.append(f"v_addc_co_u32_e32 %{H[5]}, vcc, 0, {inp('s12')}, vcc")
    # chain B: h += UE (words 0..3; word 0 of H is ue[0] itself)
    body.append(f"v_add_co_u32_e32 %{H[1]}, vcc, {inp(ue[1])}, %{H[1]}")
    body.append(f"v_addc_co_u32_e32 %{H[2]}, vcc, {inp(ue[2])}, %{H[2]}, vcc")
    body.append(f"v_addc_co_u32_e32 %{H[3]}, vcc, {inp(ue[3])}, %{H[3]}, vcc")
    body.append(f"v_addc_co_u32_e32 %{H[4]}, vcc, 0, %{H[4]}, vcc")
    body.append(f"v_addc_co_u32_e32 %{H[5]}, vcc, 0, %{H[5]}, vcc")
    # chain C: r = lo + H
    body.append(f"v_add_co_u32_e32 %{Rr[0]}, vcc, {inp('s0')}, {inp(ue[0])}")
    for k in range(1, 6):
        body.append(f"v_addc_co_u32_e32 %{Rr[k]}, vcc, {inp(f's{k}')}, %{H[k]}, vcc")
    for k in range(6, 8):
        body.append(f"v_addc_co_u32_e32 %{Rr[k]}, vcc, 0, {inp(f's{k}')}, vcc")
    body.append(f"s_mov_b64 %{OVF}, vcc")
    # canonicalise (same contract as the 16-word variant): take q = r + (2^32 + 977) if the value passed 2^256 or r >= p
    body.append(f"v_cmp_eq_u32_e32 vcc, -1, %{Rr[7]}")
    body.append(f"s_or_b64 %{OVF2}, %{OVF}, vcc")
    body.append("s_cbranch_scc0 .Lecfft_canon2_done_%=")
    body.append(f"v_add_co_u32_e32 %{Q[0]}, vcc, 0x3d1, %{Rr[0]}")
    body.append(f"v_addc_co_u32_e32 %{Q[1]}, vcc, 1, %{Rr[1]}, vcc")
    for k in range(2, 8):
        body.append(f"v_addc_co_u32_e32 %{Q[k]}, vcc, 0, %{Rr[k]}, vcc")
    body.append(f"s_or_b64 %{OVF2}, %{OVF}, vcc")
    for k in range(8):
        body.append(f"v_cndmask_b32_e64 %{Rr[k]}, %{Rr[k]}, %{Q[k]}, %{OVF2}")
    body.append(".Lecfft_canon2_done_%=:")
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
    for has_c in (False, True):
        name = "mul2_add_gfx950" if has_c else "mul2_gfx950"
        sig = "const elem& t, const elem& u, const elem& x" + (", const elem& c" if has_c else "")
        out.append(f"    // (t, u = t * 2^128 mod p) * x" + (" + c" if has_c else ""))
        out.append(f"    __device__ static inline elem {name}({sig}) {{")
        out += gen_product2(has_c)
        out += gen_combine2()
        out += gen_reduce2()
        out.append("    }")
        out.append("")
    with open(OUT, "w") as f:
        f.write("\n".join(out))
    print("wrote", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
