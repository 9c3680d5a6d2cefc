#!/usr/bin/env python3
"""Scatter gather block with JW sites per wave (state words by scalar loads in groups of GROUP rows, two SGPR sets),
self-contained for timing: accumulators and the tile loop live inside the asm.
usage: gen_scatter_var.py JW GROUP DEPTH [skiplast] > scatter_variant.inc      (scatter_bench.hip)"""
import sys
import os
Q, ROWS, ROWBYTES = int(os.environ.get('SC_Q', '21')), 128, 512
JW, GROUP, DEPTH = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
SKIPLAST = len(sys.argv) > 4 and sys.argv[4] == "skiplast"
DMA = sys.argv[-1] == "dma"          # one 1 KiB LDS-DMA piece per group start (product: 16 waves x 4 quarters = 64 KiB per tile)
VG = int(os.environ['SC_VG']) if 'SC_VG' in os.environ else {2: 128, 3: 168}.get(JW, 256)
acc = [VG - 2 * Q * (JW - jj) for jj in range(JW)]
d0 = acc[0] - 2 * DEPTH
assert d0 >= 8
GW = GROUP // 2                       # words per site and group
SETW = (JW * GW + 3) // 4 * 4
SA = 36                               # s32..s35 are reserved by the ABI, the compiler keeps s0..s31
SB = SA + SETW
assert SB + SETW <= 100, SB + SETW
TP = 100                              # stream pointer pair
sets = (SA, SB)
NG = ROWS // GROUP


def sloads(base, group_index):
    """JW x GW words of one group: contiguous in the stream at (group_index * JW * GW) words"""
    r = []
    piece = GW
    for jj in range(JW):
        reg = "s[%d:%d]" % (base + jj * GW, base + jj * GW + GW - 1)
        r.append("s_load_dwordx%d %s, s[%d:%d], 0x%x" % (piece, reg, TP, TP + 1, (group_index * JW + jj) * GW * 4))
    return r


def ds(r):
    k = r % DEPTH
    return "ds_read_b64 v[%d:%d], %%[vbase] offset:%d" % (d0 + 2 * k, d0 + 2 * k + 1, r * ROWBYTES)


o = ["s_mov_b32 vcc_lo, m0", "s_mov_b64 s[%d:%d], %%[sp]" % (TP, TP + 1)]
for r in range(d0, VG):
    o.append("v_mov_b32 v%d, 0" % r)
o += sloads(sets[0], 0)
o.append(".Ltile_%=:")
for r in range(DEPTH):
    o.append(ds(r))
pending = 0
for r in range(ROWS):
    if r % GROUP == 0:
        g = r // GROUP
        if r:
            o.append("s_set_gpr_idx_off")
        if DMA:
            o += ["s_mov_b32 m0, %d" % (65536 + g * 1024), "s_nop 0", "global_load_lds_dwordx4 %[voff], %[gsrc]"]
        o.append("s_waitcnt lgkmcnt(0)")
        if g + 1 < NG:
            o += sloads(sets[(g + 1) % 2], g + 1)
        else:                          # first group of the next tile
            o += ["s_add_u32 s%d, s%d, %d" % (TP, TP, NG * JW * GW * 4), "s_addc_u32 s%d, s%d, 0" % (TP + 1, TP + 1)]
            o += sloads(sets[0], 0)
        o.append("s_set_gpr_idx_on s%d, 0x9" % sets[g % 2])
    o.append("s_waitcnt lgkmcnt(%d)" % min(DEPTH - 1, ROWS - 1 - r))
    k = r % DEPTH
    cur = sets[(r // GROUP) % 2]
    for jj in range(JW):
        w = cur + jj * GW + (r % GROUP) // 2
        o.append(("s_pack_ll_b32_b16 m0, s%d, 0" if r % 2 == 0 else "s_lshr_b32 m0, s%d, 16") % w)
        o.append("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (acc[jj], acc[jj] + 1, acc[jj], acc[jj] + 1, d0 + 2 * k, d0 + 2 * k + 1))
    nxt = r + DEPTH
    if nxt < ROWS:
        if SKIPLAST and (r + 1) % GROUP == 0:
            pending = nxt              # issue it one row later, together with that row's own read
        else:
            o.append(ds(nxt))
            if pending:
                o.append(ds(pending))
                pending = 0
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
print("#define SC_STREAM_WORDS_PER_TILE %d" % (NG * JW * GW))
