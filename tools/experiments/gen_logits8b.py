#!/usr/bin/env python3
"""Logits inner block for 8-wave workgroups (2 waves per SIMD, 256 VGPRs), U sequences per wave, ONE set of state
words refilled in place with slack: the words are consumed in order, in chunks of CH sequences (one s_load each).
Per site two `s_waitcnt lgkmcnt(0)`:
  site start (after the q row reads): the chunks 0 .. m-2 of this site (issued at the previous site's mid point) have
      landed; then the LAST chunk of this site is requested (its registers were in use until the previous site's end);
  mid point (before the last chunk is used): nothing but that one load is outstanding; then chunks 0 .. m-2 of the
      NEXT site are requested (their registers are free by now).
usage: gen_logits8b.py U CH > logits_variant.inc     (gpridx_logits.hip -DNWAVES=8 -DNBSEQ=U)"""
import sys
Q, JT, ROWBYTES = 21, 6, 512
U, CH = int(sys.argv[1]), int(sys.argv[2])
OPTS = sys.argv[3:]                  # nop: s_nop 0 between the M0 write and the add; read2: rows fetched in pairs (ds_read2_b64)
assert U % CH == 0 and CH in (8, 16, 32)
M = U // CH                       # chunks per site
CW = CH // 2                      # dwords per chunk
W0, ACC = 4, 4 + 2 * Q
S0 = 36
TB = S0 + U // 2                  # temporaries: sptr pair (this site), saved m0, zero, next-site pointer pair
assert TB + 6 <= 96 and ACC + 2 * U + 2 <= 256, (TB, ACC + 2 * U)
o = ["s_mov_b32 s%d, m0" % (TB + 2), "s_mov_b64 s[%d:%d], %%[sptr]" % (TB, TB + 1), "s_mov_b32 s%d, 0" % (TB + 3)]


def sload(c, ptr):
    reg = "s[%d:%d]" % (S0 + c * CW, S0 + (c + 1) * CW - 1)
    return "s_load_dwordx%d %s, s[%d:%d], 0x%x" % (CW, reg, ptr, ptr + 1, c * CW * 4)


for c in range(M - 1):
    o.append(sload(c, TB))
for jj in range(JT):
    if "read2" in OPTS:
        for b in range(0, Q - 1, 2):              # offsets are 8 bits x 8 bytes: a pair per address register
            t = ACC + 2 * U + (b // 2) % 2
            o.append("v_add_u32 v%d, %d, %%[vbase]" % (t, (jj * Q + b) * ROWBYTES))
            o.append("ds_read2_b64 v[%d:%d], v%d offset0:0 offset1:64" % (W0 + 2 * b, W0 + 2 * b + 3, t))
        o.append("ds_read_b64 v[%d:%d], %%[vbase] offset:%d" % (W0 + 2 * (Q - 1), W0 + 2 * (Q - 1) + 1, (jj * Q + Q - 1) * ROWBYTES))
    else:
        for b in range(Q):
            o.append("ds_read_b64 v[%d:%d], %%[vbase] offset:%d" % (W0 + 2 * b, W0 + 2 * b + 1, (jj * Q + b) * ROWBYTES))
    o.append("s_waitcnt lgkmcnt(0)")
    o.append(sload(M - 1, TB))
    o.append("s_set_gpr_idx_on s%d, 0x2" % (TB + 3))
    for sq in range(U):
        if sq == (M - 1) * CH:
            o.append("s_waitcnt lgkmcnt(0)")
            if jj + 1 < JT:
                o += ["s_add_u32 s%d, s%d, %%[stride]" % (TB, TB), "s_addc_u32 s%d, s%d, 0" % (TB + 1, TB + 1)]
                for c in range(M - 1):
                    o.append(sload(c, TB))
        w = S0 + sq // 2
        o.append(("s_pack_ll_b32_b16 m0, s%d, 0" if sq % 2 == 0 else "s_lshr_b32 m0, s%d, 16") % w)
        if "nop" in OPTS:
            o.append("s_nop 0")
        a = ACC + 2 * sq
        o.append("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (a, a + 1, a, a + 1, W0, W0 + 1))
    o.append("s_set_gpr_idx_off")
o.append("s_mov_b32 m0, s%d" % (TB + 2))
clob = ['"memory"', '"scc"'] + ['"v%d"' % i for i in range(W0, ACC + 2 * U + 2)] + ['"s%d"' % i for i in range(36, TB + 4)]
print("#define LOGITS_BLOCK(VBASE, SPTR, STRIDE) asm volatile( \\")
for ln in o:
    print('    "%s\\n" \\' % ln)
print("    : \\")
print('    : [vbase] "v"(VBASE), [sptr] "s"(SPTR), [stride] "s"(STRIDE) \\')
print("    : %s)" % ", ".join(clob))
