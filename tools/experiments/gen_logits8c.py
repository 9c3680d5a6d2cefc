#!/usr/bin/env python3
"""Logits inner block, 8 waves, U sequences per wave, TWO banks of q row registers: the rows of site j+1 are fetched
into the other bank in the middle of site j, so a site starts without waiting for LDS; one set of state words refilled
in place as in gen_logits8b.py (chunks of CH sequences; chunks K .. m-1 are requested at the site start, chunks
0 .. K-1 of the next site after the mid-site wait in front of chunk K's first use).
usage: gen_logits8c.py U CH K > logits_variant.inc     (gpridx_logits.hip -DNWAVES=8 -DNBSEQ=U)"""
import sys
Q, JT, ROWBYTES = 21, 6, 512
U, CH, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
assert U % CH == 0 and CH in (8, 16, 32)
M = U // CH
CW = CH // 2
assert 0 < K < M
W = (4, 4 + 2 * Q)
ACC = 4 + 4 * Q
S0 = 36
TB = S0 + U // 2
assert TB + 6 <= 96 and ACC + 2 * U <= 256, (TB, ACC + 2 * U)
o = ["s_mov_b32 s%d, m0" % (TB + 2), "s_mov_b64 s[%d:%d], %%[sptr]" % (TB, TB + 1), "s_mov_b32 s%d, 0" % (TB + 3)]


def sload(c, ptr):
    reg = "s[%d:%d]" % (S0 + c * CW, S0 + (c + 1) * CW - 1)
    return "s_load_dwordx%d %s, s[%d:%d], 0x%x" % (CW, reg, ptr, ptr + 1, c * CW * 4)


def rows(jj, bank):
    return ["ds_read_b64 v[%d:%d], %%[vbase] offset:%d" % (W[bank] + 2 * b, W[bank] + 2 * b + 1, (jj * Q + b) * ROWBYTES) for b in range(Q)]


for c in range(K):
    o.append(sload(c, TB))
o += rows(0, 0)
for jj in range(JT):
    bank = jj % 2
    o.append("s_waitcnt lgkmcnt(0)")
    for c in range(K, M):
        o.append(sload(c, TB))
    o.append("s_set_gpr_idx_on s%d, 0x2" % (TB + 3))
    for sq in range(U):
        if sq == K * CH:
            o.append("s_waitcnt lgkmcnt(0)")
            if jj + 1 < JT:
                o += ["s_add_u32 s%d, s%d, %%[stride]" % (TB, TB), "s_addc_u32 s%d, s%d, 0" % (TB + 1, TB + 1)]
                for c in range(K):
                    o.append(sload(c, TB))
                o += rows(jj + 1, bank ^ 1)
        w = S0 + sq // 2
        o.append(("s_pack_ll_b32_b16 m0, s%d, 0" if sq % 2 == 0 else "s_lshr_b32 m0, s%d, 16") % w)
        a = ACC + 2 * sq
        o.append("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (a, a + 1, a, a + 1, W[bank], W[bank] + 1))
    o.append("s_set_gpr_idx_off")
o.append("s_mov_b32 m0, s%d" % (TB + 2))
clob = ['"memory"', '"scc"'] + ['"v%d"' % i for i in range(4, ACC + 2 * U)] + ['"s%d"' % i for i in range(36, TB + 4)]
print("#define LOGITS_BLOCK(VBASE, SPTR, STRIDE) asm volatile( \\")
for ln in o:
    print('    "%s\\n" \\' % ln)
print("    : \\")
print('    : [vbase] "v"(VBASE), [sptr] "s"(SPTR), [stride] "s"(STRIDE) \\')
print("    : %s)" % ", ".join(clob))
