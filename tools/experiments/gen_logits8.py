#!/usr/bin/env python3
"""Logits inner block for 8-wave workgroups (2 waves per SIMD, 256 VGPRs): U sequences per wave, one bank of q row
registers, state words in two half-sets that are refilled as soon as their half of the site is done.
usage: gen_logits8.py U [nop] > logits_variant.inc     (gpridx_logits.hip -DNWAVES=8 -DNBSEQ=U)"""
import sys
Q, JT, ROWBYTES = 21, 6, 512
U = int(sys.argv[1])
NOP = len(sys.argv) > 2 and sys.argv[2] == "nop"
assert U % 16 == 0
HW = U // 4                       # state words per half-set
W0, ACC = 4, 4 + 2 * Q
SA, SB = 36, 36 + (HW + 3) // 4 * 4
TB = SB + (HW + 3) // 4 * 4       # temporaries: sptr pair, saved m0, zero
assert TB + 4 <= 100 and ACC + 2 * U <= 256
o = ["s_mov_b32 s%d, m0" % (TB + 2), "s_mov_b64 s[%d:%d], %%[sptr]" % (TB, TB + 1), "s_mov_b32 s%d, 0" % (TB + 3)]


def sloads(base, off):
    r, left, at = [], HW, 0
    for piece in (16, 8, 4, 2, 1):
        while left >= piece:
            reg = "s[%d:%d]" % (base + at, base + at + piece - 1) if piece > 1 else "s%d" % (base + at)
            r.append("s_load_dword%s %s, s[%d:%d], 0x%x" % ("x%d" % piece if piece > 1 else "", reg, TB, TB + 1, off + at * 4))
            at += piece; left -= piece
    return r


o += sloads(SA, 0) + sloads(SB, HW * 4)
for jj in range(JT):
    for b in range(Q):
        o.append("ds_read_b64 v[%d:%d], %%[vbase] offset:%d" % (W0 + 2 * b, W0 + 2 * b + 1, (jj * Q + b) * ROWBYTES))
    o.append("s_waitcnt lgkmcnt(0)")
    if jj + 1 < JT:
        o += ["s_add_u32 s%d, s%d, %%[stride]" % (TB, TB), "s_addc_u32 s%d, s%d, 0" % (TB + 1, TB + 1)]
    o.append("s_set_gpr_idx_on s%d, 0x2" % (TB + 3))
    for sq in range(U):
        base = SA if sq < U // 2 else SB
        w = base + (sq % (U // 2)) // 2
        o.append(("s_pack_ll_b32_b16 m0, s%d, 0" if sq % 2 == 0 else "s_lshr_b32 m0, s%d, 16") % w)
        if NOP:
            o.append("s_nop 0")
        a = ACC + 2 * sq
        o.append("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (a, a + 1, a, a + 1, W0, W0 + 1))
        if sq == U // 2 - 1 and jj + 1 < JT:
            o += sloads(SA, 0)
    o.append("s_set_gpr_idx_off")
    if jj + 1 < JT:
        o += sloads(SB, HW * 4)
o.append("s_mov_b32 m0, s%d" % (TB + 2))
clob = ['"memory"', '"scc"'] + ['"v%d"' % i for i in range(W0, ACC + 2 * U)] + ['"s%d"' % i for i in range(36, TB + 4)]
print("#define LOGITS_BLOCK(VBASE, SPTR, STRIDE) asm volatile( \\")
for ln in o:
    print('    "%s\\n" \\' % ln)
print("    : \\")
print('    : [vbase] "v"(VBASE), [sptr] "s"(SPTR), [stride] "s"(STRIDE) \\')
print("    : %s)" % ", ".join(clob))
