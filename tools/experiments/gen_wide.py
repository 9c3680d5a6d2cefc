#!/usr/bin/env python3
"""Gather block with 16-byte lanes (256 columns per wave), two sites per wave, 256 VGPRs
(2 waves per SIMD): one M0 write serves two packed adds.  Timing experiment only.
usage: gen_wide.py > gather_wide.inc"""
ROWS, ROWBYTES, JW, DEPTH, Q = 64, 1024, 2, 8, 21
S0, T0 = 40, 72
A = [88, 88 + 4 * Q]
D0 = 88 - 4 * DEPTH
o = ["s_mov_b32 s%d, m0" % (T0 + 1)]


def ds(r):
    k = r % DEPTH
    return "ds_read_b128 v[%d:%d], %%[vbase] offset:%d" % (D0 + 4 * k, D0 + 4 * k + 3, r * ROWBYTES)


for r in range(DEPTH):
    o.append(ds(r))
quarter = ROWS // 2
for r in range(ROWS):
    if r % quarter == 0:
        if r:
            o.append("s_set_gpr_idx_off")
        for jj in range(JW):
            for w in range(16):
                o.append("v_readlane_b32 s%d, %%[st%d], %d" % (S0 + jj * 16 + w, jj, (r // quarter) * 16 + w))
        o.append("s_nop 3")
        o.append("s_mov_b32 s%d, 0" % T0)
        o.append("s_set_gpr_idx_on s%d, 0x9" % T0)
    o.append("s_waitcnt lgkmcnt(%d)" % min(DEPTH - 1, ROWS - 1 - r))
    k = r % DEPTH
    for jj in range(JW):
        w = S0 + jj * 16 + (r % quarter) // 2
        o.append(("s_pack_ll_b32_b16 m0, s%d, 0" if r % 2 == 0 else "s_lshr_b32 m0, s%d, 16") % w)
        for h in range(2):
            o.append("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (A[jj] + 2 * h, A[jj] + 2 * h + 1, A[jj] + 2 * h, A[jj] + 2 * h + 1,
                                                                   D0 + 4 * k + 2 * h, D0 + 4 * k + 2 * h + 1))
    if r + DEPTH < ROWS:
        o.append(ds(r + DEPTH))
o.append("s_set_gpr_idx_off")
o.append("s_mov_b32 m0, s%d" % (T0 + 1))
clob = ['"memory"'] + ['"v%d"' % i for i in range(D0, 256)] + ['"s%d"' % (S0 + i) for i in range(32)] + ['"s72"', '"s73"']
print("#define GATHER_WIDE(VBASE, ST0, ST1) asm volatile( \\")
for ln in o:
    print('    "%s\\n" \\' % ln)
print("    : \\")
print('    : [vbase] "v"(VBASE), [st0] "v"(ST0), [st1] "v"(ST1) \\')
print("    : %s)" % ", ".join(clob))
