#!/usr/bin/env python3
"""Scatter gather block, block-pipelined: rows and state words of block b+1 (BLK rows x JW sites) are requested at the
start of block b -- BLK ds_read_b64 into the other half of the row ring, JW scalar loads into the other SGPR set -- and
the only wait is ONE `s_waitcnt lgkmcnt(0)` per block, a whole block of adds after the requests (no counted per-row
waits, which over-count while scalar loads are in flight).  Self-contained for timing as gen_scatter_var.py.
usage: gen_scatter_blk.py JW BLK > scatter_variant.inc      (scatter_bench.hip -DNWAVES=...)"""
import sys
import os
Q, ROWS, ROWBYTES = int(os.environ.get('SC_Q', '21')), 128, 512
JW, BLK = int(sys.argv[1]), int(sys.argv[2])
DMA = len(sys.argv) > 3 and sys.argv[3] == "dma"
WAVES = int(sys.argv[4]) if len(sys.argv) > 4 else 8
PPW = 64 // WAVES                     # 1 KiB LDS-DMA pieces per wave and tile
VG = int(os.environ['SC_VG']) if 'SC_VG' in os.environ else {2: 128, 3: 168}.get(JW, 256)
acc = [VG - 2 * Q * (JW - jj) for jj in range(JW)]
d0 = acc[0] - 4 * BLK                 # two ring halves of BLK register pairs
assert d0 >= 8, d0
GW = BLK // 2                         # words per site and block
assert GW in (2, 4, 8, 16)
SETW = (JW * GW + 3) // 4 * 4
SA = 36
SB = SA + SETW
assert SB + SETW <= 100, SB + SETW
TP = 100
sets = (SA, SB)
NB = ROWS // BLK


def sloads(base, block_index):
    r = []
    for jj in range(JW):
        reg = "s[%d:%d]" % (base + jj * GW, base + jj * GW + GW - 1)
        r.append("s_load_dwordx%d %s, s[%d:%d], 0x%x" % (GW, reg, TP, TP + 1, (block_index * JW + jj) * GW * 4))
    return r


def row_reg(b, i):
    return d0 + 2 * ((b % 2) * BLK + i)


def dsreads(b):
    return ["ds_read_b64 v[%d:%d], %%[vbase] offset:%d" % (row_reg(b, i), row_reg(b, i) + 1, ((b % NB) * BLK + i) * ROWBYTES) for i in range(BLK)]


o = ["s_mov_b32 vcc_lo, m0", "s_mov_b64 s[%d:%d], %%[sp]" % (TP, TP + 1)]
for r in range(d0, VG):
    o.append("v_mov_b32 v%d, 0" % r)
o += sloads(sets[0], 0) + dsreads(0)
o.append(".Ltile_%=:")
assert NB % 2 == 0
for b in range(NB):
    if DMA and b % (NB // PPW) == 0:
        o += ["s_mov_b32 m0, %d" % (65536 + (b // (NB // PPW)) * 1024), "s_nop 0", "global_load_lds_dwordx4 %[voff], %[gsrc]"]
    o.append("s_waitcnt lgkmcnt(0)")
    if b + 1 < NB:
        o += sloads(sets[(b + 1) % 2], b + 1)
    else:
        o += ["s_add_u32 s%d, s%d, %d" % (TP, TP, NB * JW * GW * 4), "s_addc_u32 s%d, s%d, 0" % (TP + 1, TP + 1)]
        o += sloads(sets[0], 0)
    o += dsreads(b + 1)
    o.append("s_set_gpr_idx_on s%d, 0x9" % sets[b % 2])
    cur = sets[b % 2]
    for i in range(BLK):
        for jj in range(JW):
            w = cur + jj * GW + i // 2
            o.append(("s_pack_ll_b32_b16 m0, s%d, 0" if i % 2 == 0 else "s_lshr_b32 m0, s%d, 16") % w)
            o.append("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (acc[jj], acc[jj] + 1, acc[jj], acc[jj] + 1, row_reg(b, i), row_reg(b, i) + 1))
    o.append("s_set_gpr_idx_off")
o += ["s_sub_u32 %[iters], %[iters], 1", "s_cmp_lg_u32 %[iters], 0", "s_cbranch_scc1 .Ltile_%="]
o.append("s_waitcnt lgkmcnt(0)")
o.append("s_mov_b32 m0, vcc_lo")
o.append("v_mov_b32 %%[res], v%d" % acc[0])
print("#define SC_JW %d" % JW)
print("#define SC_VGPRS %d" % VG)
print("#define SCATTER_BLOCK(VBASE, SP, ITERS, RES%s) asm volatile( \\" % (", VOFF, GSRC" if DMA else ""))
for ln in o:
    print('    "%s\\n" \\' % ln)
print('    : [res] "=v"(RES), [iters] "+s"(ITERS) \\')
print('    : [vbase] "v"(VBASE), [sp] "s"(SP)%s \\' % (', [voff] "v"(VOFF), [gsrc] "s"(GSRC)' if DMA else ""))
clob = ['"memory"', '"scc"', '"vcc"'] + ['"v%d"' % i for i in range(d0, VG)] + ['"s%d"' % i for i in range(SA, SB + SETW)] + ['"s100"', '"s101"']
print("    : %s)" % ", ".join(clob))
print("#define SC_STREAM_WORDS_PER_TILE %d" % (NB * JW * GW))
