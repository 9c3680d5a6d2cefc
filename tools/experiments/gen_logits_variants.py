#!/usr/bin/env python3
"""Logits inner block variants for gpridx_logits.hip (timing only).
  single : product form -- one bank of q row registers, lgkmcnt(0) after the q reads of every site
  double : two banks (2 x 2q registers): the rows and state words of site j+1 are fetched while
           site j is being added; needs 168 VGPRs (3 waves per SIMD, 12-wave workgroups)
usage: gen_logits_variants.py single|double > logits_variant.inc"""
import sys
Q, JT, NBSEQ, ROWBYTES = 21, 6, 32, 512
variant = sys.argv[1]
B128 = variant == "wide128"      # row pairs fetched by ds_read_b128 (tile rows interleaved in pairs)
if B128:
    variant = "wide"
if variant == "wide":
    # wide N: single bank, N sequences per wave (N/2 state SGPRs per site, two ping-pong sets)
    NBSEQ = int(sys.argv[2])
    ACC, W = 4 + 2 * Q, [4]
    NS = (NBSEQ // 2 + 3) // 4 * 4
    S = (36, 36 + NS)
    assert 36 + 2 * NS <= 100
elif variant == "single":
    ACC, W = 64, [64 - 2 * Q]
else:
    ACC, W = 104, [20, 62]
if variant != "wide":
    S = (40, 56)
o = ["s_mov_b32 vcc_lo, m0", "s_mov_b64 s[74:75], %[sptr]"]


def loads(jj, bank, sset):
    if variant == "wide":
        r, left, at = [], NBSEQ // 2, 0
        for piece in (16, 8, 4, 2, 1):
            while left >= piece:
                nm = "s_load_dword" + ("x%d" % piece if piece > 1 else "")
                reg = "s[%d:%d]" % (S[sset] + at, S[sset] + at + piece - 1) if piece > 1 else "s%d" % (S[sset] + at)
                r.append("%s %s, s[34:35], 0x%x" % (nm, reg, at * 4))
                at += piece; left -= piece
        r += ["s_add_u32 s34, s34, %[stride]", "s_addc_u32 s35, s35, 0"]
    else:
        r = ["s_load_dwordx16 s[%d:%d], s[74:75], 0x0" % (S[sset], S[sset] + 15), "s_add_u32 s74, s74, %[stride]", "s_addc_u32 s75, s75, 0"]
    if B128:
        for b in range(0, Q - 1, 2):
            r.append("ds_read_b128 v[%d:%d], %%[vbase2] offset:%d" % (W[bank] + 2 * b, W[bank] + 2 * b + 3, (jj * Q + b) * ROWBYTES))
        r.append("ds_read_b64 v[%d:%d], %%[vbase] offset:%d" % (W[bank] + 2 * Q - 2, W[bank] + 2 * Q - 1, (jj * Q + Q - 1) * ROWBYTES))
        return r
    for b in range(Q):
        r.append("ds_read_b64 v[%d:%d], %%[vbase] offset:%d" % (W[bank] + 2 * b, W[bank] + 2 * b + 1, (jj * Q + b) * ROWBYTES))
    return r


def units(bank, sset):
    r = ["s_set_gpr_idx_on s%d, 0x2" % S[sset]]
    for sq in range(NBSEQ):
        w = S[sset] + sq // 2
        r.append(("s_pack_ll_b32_b16 m0, s%d, 0" if sq % 2 == 0 else "s_lshr_b32 m0, s%d, 16") % w)
        a = ACC + 2 * sq
        import os
        if os.environ.get("LG_NOP"):
            r.append("s_nop %s" % os.environ["LG_NOP"])
        r.append("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (a, a + 1, a, a + 1, W[bank], W[bank] + 1))
    r.append("s_set_gpr_idx_off")
    return r


if variant == "wide":
    o[1] = "s_mov_b64 s[34:35], %[sptr]"
    o += [x for x in loads(0, 0, 0) if x.startswith("s_")]
    for jj in range(JT):
        o += [x for x in loads(jj, 0, 0) if x.startswith("ds_")]
        o.append("s_waitcnt lgkmcnt(0)")
        if jj + 1 < JT:
            o += [x for x in loads(jj + 1, 0, (jj + 1) % 2) if x.startswith("s_")]
        o += units(0, jj % 2)
elif variant == "single":
    for jj in range(JT):
        o += loads(jj, 0, jj % 2)
        o.append("s_waitcnt lgkmcnt(0)")
        o += units(0, jj % 2)
else:
    o += loads(0, 0, 0)
    for jj in range(JT):
        o.append("s_waitcnt lgkmcnt(0)")
        if jj + 1 < JT:
            o += loads(jj + 1, (jj + 1) % 2, (jj + 1) % 2)
        o += units(jj % 2, jj % 2)
o.append("s_mov_b32 m0, vcc_lo")
lo = min(W)
clob = ['"memory"', '"vcc"'] + ['"v%d"' % i for i in range(lo, ACC + 2 * NBSEQ)] + ['"s%d"' % i for i in (range(34, 100) if variant == "wide" else range(40, 76))]
print("#define LOGITS_BLOCK(VBASE, SPTR, STRIDE) asm volatile( \\")
for ln in o:
    print('    "%s\\n" \\' % ln)
print("    : \\")
print('    : [vbase] "v"(VBASE), [vbase2] "v"(2 * (VBASE) - (uint32_t)(uintptr_t)smem), [sptr] "s"(SPTR), [stride] "s"(STRIDE) \\')
print("    : %s)" % ", ".join(clob))
