#!/usr/bin/env python3
"""Generates the inline-asm inner blocks of the two gather kernels of plm_engine.hip, one macro per
(q, element type): pydca_amd/csrc/scatter_gather_asm.inc (plm_scatter_kernel, described first) and
pydca_amd/csrc/logits_gather_asm.inc (plm_logits_kernel, described at logits_body below).

One block = one wave, one 128-row x 512-byte LDS tile of R, JW = 2 sites:

    for r in 0..127:   v = tile[r][lane]  (8 bytes: two floats or one double)
                       acc[site0][state(site0, r)] += v ;  acc[site1][state(site1, r)] += v

The accumulator of a state is selected with the gfx9 VGPR index mode (s_set_gpr_idx_on /
s_set_gpr_idx_idx: M0[7:0] is added to the VGPR number of src0 and dst), so rows are visited
in sequence order: no sorted lists, no per-state loops, no padding, and the LDS address of row
r is an immediate offset.  Per row: 1 ds_read_b64 + per site one SALU write of M0 and one
v_pk_add_f32 | v_add_f64.  Eight reads are kept in flight (data ring of 8 register pairs).

The CU's single scalar ALU issues one instruction per clock, so the M0 update must be ONE SALU
instruction per (row, site): the state array holds ready-made 16-bit M0 images
(0x9000 | 2 x state: index-enable bits for src0 and dst + register-pair offset), two rows per
dword, and `s_pack_ll_b32_b16 m0, w, 0` / `s_lshr_b32 m0, w, 16` move them into place.  (The
first version used s_bfe_u32 + s_set_gpr_idx_idx per (row, site) and measured SALU-bound at
3 clk per (row, site) per CU.)  The 128 entries of a site arrive in one VGPR (lane l holds rows
2l, 2l+1) and are moved to SGPRs with v_readlane, 16 words per quarter tile.

Register plan (physical, fixed; the kernel is built for 128 VGPRs = 4 waves per SIMD):
    q=21: data ring v[28:43], site 0 v[44:85], site 1 v[86:127]
    q=5 : data ring v[92:107], site 0 v[108:117], site 1 v[118:127]
    q=5, five sites per wave (_JW5 macros): data ring v[62:77], sites v[78:127], 8 state words per
          site at a time (16 rows per refill) so that 40 SGPRs hold them
    state words s[40:71] (s[40:79] with five sites), temporaries behind them: s72 / s80 (zero), s73 / s81 (saved M0)
Accumulator tuples are passed as "+{v[a:b]}" operands so the compiler knows they live there.

Checked on MI355X with tools/experiments/gpridx_bench.hip: bit-exact against sequential CPU
sums, ~100 TB/s aggregate LDS read rate for the single-site form.
"""
import os

ROWS, ROWBYTES, DEPTH, JW = 128, 512, 8, 2
STAGE_AT = os.environ.get("DCA_GEN_STAGE_AT", "pre")
SC_DMA_AUX = os.environ.get("DCA_GEN_SC_DMA_AUX", "")       # cache-policy bits of the scatter / logits LDS-DMA loads (experiments: " nt", " sc1" ...)
LG_DMA_AUX = os.environ.get("DCA_GEN_LG_DMA_AUX", "")
SC_DMA_PLAN = [int(c) for c in os.environ.get("DCA_GEN_SC_DMA_PLAN", "1111")]      # LDS-DMA pieces of the next tile issued per quarter
# timing-only ablations of the scatter block (WRONG results): 1 no per-row lgkmcnt waits, 2 no quarter-boundary drain and
# no state-word refresh after the first quarter, 4 no LDS row reads, 8 no M0 writes
SC_EXP = int(os.environ.get("DCA_GEN_SC_EXP", "0"))
assert len(SC_DMA_PLAN) == 4 and sum(SC_DMA_PLAN) == 4
S0, T0 = 40, 72


def tuples(q):
    """legal VGPR tuple sizes covering 2q registers"""
    return {21: [32, 8, 2], 5: [8, 2], 25: [32, 16, 2]}[q]


def plan(q, jw):
    """-> (first data-ring register, [first accumulator register of every site])"""
    acc = [128 - 2 * q * (jw - jj) for jj in range(jw)]
    return acc[0] - 2 * DEPTH, acc


def group_words(jw):
    """state words per site held in SGPRs at a time (two rows per word): 32 SGPRs for two sites,
    40 for five."""
    return 16 if jw <= 2 else 8


def temp_base(jw):
    """first of the two temporary SGPRs (zero / index, saved M0): behind the state words"""
    return max(T0, S0 + group_words(jw) * jw)


SET_A, SET_B = 36, 68          # two sets of 2 x 16 state words (scalar-load variant, two sites per wave)


# LDS-DMA pieces of the next tile per wave and quarter tile, by workgroup size: 64 pieces per tile over 16 / 8 / 4 waves
SC_PLANS = {16: SC_DMA_PLAN, 8: [2, 2, 2, 2], 4: [4, 4, 4, 4]}
# How the next tile of R reaches LDS in the 16-wave scatter block.  "dma": global_load_lds_dwordx4 (no staging registers).
# "vgpr" (round 4): global_load_dwordx4 into four VGPRs at the start of a quarter tile, ds_write_b128 at the start of the next:
# the LDS-DMA path fills LDS at ~12 B/clk/CU, and the scatter kernel needs 64 KiB per ~6100-clk tile (10.7 B/clk/CU) -- it
# looked as if it ran at the DMA's ceiling as much as at the adds' (dropping the staging altogether: 7.0 -> 5.9 ms at config D).
# MEASURED (profiles/r04_scatter_vgpr_staging.txt): slower -- D 7.62 against 7.04 ms, E 0.78 against 0.71, float64 D 15.5 against
# 13.9: the register path pays a vmcnt wait and a 1 KiB LDS write per piece in the instruction stream of waves that have no
# slack, and the fill rate was not the limit.  The product ships "dma"; the option stays for the record.
SC_STAGE = os.environ.get("DCA_GEN_SC_STAGE", "dma")


def body_smem(q, f64, waves=16):
    """Two sites per wave, state words through scalar loads.  Moving the words to SGPRs with v_readlane
    costs 0.5 VALU instruction per (row, site) and the VALU is the busiest unit of this block; an
    s_load_dwordx16 per site and quarter tile costs none.  Quarter k uses SGPR set k % 2; the loads of
    quarter k+1 are issued at the start of quarter k and the quarter boundary waits lgkmcnt(0) (scalar
    loads return out of order, so this is the only safe wait; it also drains the eight LDS reads in
    flight -- a bubble of one LDS latency per 32 rows).  While a scalar load is in flight the per-row
    `s_waitcnt lgkmcnt(7)` over-counts by up to two, which only makes it wait longer.  Measured in
    isolation (tools/experiments/gen_variants.py smem): 1.70 clk per (row, site) per CU against 2.15
    for the v_readlane form.  The saved M0 lives in vcc_lo."""
    d0, acc = plan(q, 2)
    add = "v_add_f64" if f64 else "v_pk_add_f32"
    sets = (SET_A, SET_B)
    o = ["s_mov_b32 vcc_lo, m0"]

    def ds(r):
        k = r % DEPTH
        return "ds_read_b64 v[%d:%d], %%[vbase] offset:%d" % (d0 + 2 * k, d0 + 2 * k + 1, r * ROWBYTES)

    def sload(qk, base):
        return ["s_load_dwordx16 s[%d:%d], %%[sp%d], 0x%x" % (base + 16 * jj, base + 16 * jj + 15, jj, qk * 64) for jj in range(2)]

    vstage = SC_STAGE == "vgpr" and waves == 16 and SC_PLANS[waves] == [1, 1, 1, 1]
    o += sload(0, sets[0])
    o.append("v_mov_b32 %[vtmp], %[voff]")
    if vstage:      # LDS address of this lane's 16 bytes of the wave's first piece of the next tile
        o += ["v_mbcnt_lo_u32_b32 %[vw], -1, 0", "v_mbcnt_hi_u32_b32 %[vw], -1, %[vw]", "v_lshlrev_b32 %[vw], 4, %[vw]",
              "v_add_u32 %[vw], %[ldst], %[vw]"]
    for r in range(DEPTH):
        o.append(ds(r))
    quarter = ROWS // 4
    for r in range(ROWS):
        if r % quarter == 0:
            qk = r // quarter
            if r:
                o.append("s_set_gpr_idx_off")
            # this wave's LDS-DMA piece qk of the NEXT tile (4 pieces per wave and tile), issued here rather
            # than all 64 pieces of the workgroup at the tile start: no burst of LDS writes in front of everyone's reads
            # SC_DMA_PLAN[qk] pieces are issued at the start of quarter qk (default one per quarter)
            dplan = SC_PLANS[waves]
            first = sum(dplan[:qk])
            dma = []
            if vstage:
                dma += ["s_cmp_lg_u32 %[npc], 0", "s_cbranch_scc0 .Ldca_sc_skip%d_%%=" % qk]
                if qk:
                    dma += ["s_waitcnt vmcnt(0)", "ds_write_b128 %%[vw], %%[stg] offset:%d" % ((qk - 1) * 1024)]
                dma += ["global_load_dwordx4 %[stg], %[vtmp], %[gbase]" + SC_DMA_AUX, "v_add_u32 %[vtmp], %[ginc], %[vtmp]",
                        ".Ldca_sc_skip%d_%%=:" % qk]
            for piece in range(first, first + dplan[qk]) if not vstage else ():
                dma += ["s_cmp_lg_u32 %[npc], 0",
                        "s_cbranch_scc0 .Ldca_sc_skip%d_%%=" % piece,
                        "s_add_u32 m0, %%[ldst], %d" % (piece * 1024),
                        "s_nop 0",
                        "global_load_lds_dwordx4 %[vtmp], %[gbase]" + SC_DMA_AUX,
                        "v_add_u32 %[vtmp], %[ginc], %[vtmp]",
                        ".Ldca_sc_skip%d_%%=:" % piece]
            if STAGE_AT == "pre":
                o += dma
            if not (SC_EXP & 2) or qk == 0:
                o.append("s_waitcnt lgkmcnt(0)")
            if qk + 1 < 4 and not (SC_EXP & 2):
                o += sload(qk + 1, sets[(qk + 1) % 2])
            if STAGE_AT == "post":
                o += dma
            o.append("s_set_gpr_idx_on s%d, 0x9" % sets[qk % 2])       # index bits are rewritten before every use
        if STAGE_AT == "mid" and r % quarter == quarter // 2:
            qk = r // quarter
            o.append("s_set_gpr_idx_off")
            o += dma
            o.append("s_set_gpr_idx_on s%d, 0x9" % sets[qk % 2])
        if not (SC_EXP & 1):
            o.append("s_waitcnt lgkmcnt(%d)" % min(DEPTH - 1, ROWS - 1 - r))
        k = r % DEPTH
        cur = sets[0 if (SC_EXP & 2) else (r // quarter) % 2]
        for jj in range(2):
            w = cur + jj * 16 + (r % quarter) // 2
            if not (SC_EXP & 8):
                if r % 2 == 0:
                    o.append("s_pack_ll_b32_b16 m0, s%d, 0" % w)
                else:
                    o.append("s_lshr_b32 m0, s%d, 16" % w)
            o.append("%s v[%d:%d], v[%d:%d], v[%d:%d]" % (add, acc[jj], acc[jj] + 1, acc[jj], acc[jj] + 1,
                                                         d0 + 2 * k, d0 + 2 * k + 1))
        if r + DEPTH < ROWS and not (SC_EXP & 4):
            o.append(ds(r + DEPTH))
    if SC_EXP:
        o.append("s_waitcnt lgkmcnt(0)")
    o.append("s_set_gpr_idx_off")
    if vstage:      # the last piece; everything this wave stages is in LDS when the block ends (the tile barrier follows)
        o += ["s_cmp_lg_u32 %[npc], 0", "s_cbranch_scc0 .Ldca_sc_skipend_%=", "s_waitcnt vmcnt(0)",
              "ds_write_b128 %[vw], %[stg] offset:3072", "s_waitcnt lgkmcnt(0)", ".Ldca_sc_skipend_%=:"]
    o.append("s_mov_b32 m0, vcc_lo")
    return o


def macro_smem(q, f64, waves=16):
    d0, acc = plan(q, 2)
    names = "ABC"[:len(tuples(q))]
    params = ["VBASE", "SP0", "SP1", "NPC", "GBASE", "GINC", "VOFF", "LDST", "VTMP", "STG", "VW"] + ["%s%d" % (n, jj) for jj in range(2) for n in names]
    suffix = "" if waves == 16 else "_W%d" % waves
    lines = ["#define DCA_GATHER_Q%d_%s_SMEM%s(%s) \\" % (q, "F64" if f64 else "F32", suffix, ", ".join(params)), "    asm volatile( \\"]
    for ln in body_smem(q, f64, waves):
        lines.append('        "%s\\n" \\' % ln)
    outs = []
    for jj, base in enumerate(acc):
        r = base
        for n, sz in zip(names, tuples(q)):
            outs.append('"+{v[%d:%d]}"(%s%d)' % (r, r + sz - 1, n, jj))
            r += sz
    outs.append('[vtmp] "=&v"(VTMP)')
    if SC_STAGE == "vgpr" and waves == 16 and SC_PLANS[waves] == [1, 1, 1, 1]:
        outs += ['[stg] "=&v"(STG)', '[vw] "=&v"(VW)']
    lines.append("        : %s \\" % ", ".join(outs))
    lines.append('        : [vbase] "v"(VBASE), [sp0] "s"(SP0), [sp1] "s"(SP1), [npc] "s"(NPC), [gbase] "s"(GBASE), [ginc] "s"(GINC), '
                 '[voff] "v"(VOFF), [ldst] "s"(LDST) \\')
    clob = ['"memory"', '"vcc"', '"scc"'] + ['"v%d"' % (d0 + i) for i in range(2 * DEPTH)] + ['"s%d"' % i for i in range(SET_A, SET_B + 32)]
    lines.append("        : %s)" % ", ".join(clob))
    return "\n".join(lines)


def body(q, f64, jw=JW):
    d0, acc = plan(q, jw)
    gw = group_words(jw)
    T0 = temp_base(jw)
    add = "v_add_f64" if f64 else "v_pk_add_f32"
    o = ["s_mov_b32 s%d, m0" % (T0 + 1)]

    def ds(r):
        k = r % DEPTH
        return "ds_read_b64 v[%d:%d], %%[vbase] offset:%d" % (d0 + 2 * k, d0 + 2 * k + 1, r * ROWBYTES)

    for r in range(DEPTH):
        o.append(ds(r))
    group = 2 * gw                     # rows per SGPR refill
    for r in range(ROWS):
        if r % group == 0:
            if r:
                o.append("s_set_gpr_idx_off")
            for jj in range(jw):
                for w in range(gw):
                    o.append("v_readlane_b32 s%d, %%[st%d], %d" % (S0 + jj * gw + w, jj, (r // group) * gw + w))
            o.append("s_nop 3")
            o.append("s_mov_b32 s%d, 0" % T0)
            o.append("s_set_gpr_idx_on s%d, 0x9" % T0)
        o.append("s_waitcnt lgkmcnt(%d)" % min(DEPTH - 1, ROWS - 1 - r))
        k = r % DEPTH
        for jj in range(jw):
            w = S0 + jj * gw + (r % group) // 2
            if r % 2 == 0:
                o.append("s_pack_ll_b32_b16 m0, s%d, 0" % w)
            else:
                o.append("s_lshr_b32 m0, s%d, 16" % w)
            o.append("%s v[%d:%d], v[%d:%d], v[%d:%d]" % (add, acc[jj], acc[jj] + 1, acc[jj], acc[jj] + 1,
                                                         d0 + 2 * k, d0 + 2 * k + 1))
        if r + DEPTH < ROWS:
            o.append(ds(r + DEPTH))
    o.append("s_set_gpr_idx_off")
    o.append("s_mov_b32 m0, s%d" % (T0 + 1))
    return o


def macro(q, f64, jw=JW):
    d0, acc = plan(q, jw)
    gw = group_words(jw)
    T0 = temp_base(jw)
    names = "ABC"[:len(tuples(q))]
    params = ["VBASE"] + ["ST%d" % jj for jj in range(jw)] + ["%s%d" % (n, jj) for jj in range(jw) for n in names]
    suffix = "" if jw == JW else "_JW%d" % jw
    lines = ["#define DCA_GATHER_Q%d_%s%s(%s) \\" % (q, "F64" if f64 else "F32", suffix, ", ".join(params)), "    asm volatile( \\"]
    for ln in body(q, f64, jw):
        lines.append('        "%s\\n" \\' % ln)
    outs = []
    for jj, base in enumerate(acc):
        r = base
        for n, sz in zip(names, tuples(q)):
            outs.append('"+{v[%d:%d]}"(%s%d)' % (r, r + sz - 1, n, jj))
            r += sz
    lines.append("        : %s \\" % ", ".join(outs))
    lines.append("        : [vbase] \"v\"(VBASE), %s \\" % ", ".join('[st%d] "v"(ST%d)' % (jj, jj) for jj in range(jw)))
    clob = ['"memory"'] + ['"v%d"' % (d0 + i) for i in range(2 * DEPTH)] + \
           ['"s%d"' % (S0 + i) for i in range(gw * jw)] + ['"s%d"' % T0, '"s%d"' % (T0 + 1)]
    lines.append("        : %s)" % ", ".join(clob))
    return "\n".join(lines)


# ------------------------------------------------------------------------------------------ logits
# One block = one wave, one LDS tile of JT sites, NS sequences (LOGITS_CFG), one 512-byte column strip.
# For every site j of the tile:
#
#     w[b] = Wtile[(j, b)][lane]  for b in 0..q-1      (q ds_read_b64 with immediate offsets)
#     for s in 0..NS-1:  acc[s] += w[state(n0 + s, j)] (source register selected through M0)
#
# i.e. the transpose of the scatter block: there the destination is indexed, here the source
# (src1 relative: M0 image 0x2000 | 2 x state).  The q rows of W of the site sit in q register
# pairs, so a (sequence, site) pair costs one SALU write of M0 and one packed add and no LDS access
# of its own; the LDS reads are q per NS pairs.  The NS M0 images of a site arrive by scalar loads
# issued one site ahead (two SGPR sets, ping-pong), so only the LDS latency of the q row reads is
# exposed per site -- the other waves of the SIMD cover it.
#
# Register plan (logits_plan): accumulators at the top of the VGPR budget (128 for 16-wave, 168 for
# 12-wave workgroups) as 16-register tuples, the q row pairs right below them; state words from s36
# in two sets, behind them the temporaries: zero, saved M0, state pointer (2), staging source (2) and
# LDS destination.


PAIR_JT = 12                   # pair variant: site pairs per LDS tile of the macro being generated (main() also emits 11 and 10)


def logits_jt(q):
    return {21: 6, 5: 25, 25: PAIR_JT}[q]     # q = 25: PAIRS of q = 5 sites per tile (24 sites, 120 rows)


# (waves per workgroup, sequences per wave) of plm_logits_kernel.  Per site a wave spends a fixed ~490 clk fetching
# the q rows and ~15.5 clk per sequence (the M0 write -> indexed add chain), so more sequences per wave amortise the
# fetch, and more sequences per workgroup share one staged tile of W.  q=21: 8 waves x 96 sequences on 256 VGPRs
# (2 waves per SIMD).  The 48 state dwords of a site then fit the SGPR budget (s36..s95) only ONCE, so the single set
# is refilled in place, in thirds, each request >= a third of a site ahead of its use (logits_body); measured in
# isolation with L2-hot state words (tools/experiments/gen_logits8b.py): 1.66 clk per (sequence, site) per CU against
# 1.88 for 12 waves x 56 sequences with two ping-pong sets (round 1's shape; 16 x 32: 2.25).
# q=5: 16 waves x 48 sequences, two sets.
# "q = 25" is the site-pair alphabet of q = 5 in float32 (round 5, logits_body): a unit of the block is a PAIR of sites, its
# 25 "rows" are the sums W[(j1, b1)] + W[(j2, b2)] formed once per wave and pair in registers (10 row reads + 25 packed adds),
# and a sequence then costs ONE M0 write and ONE indexed add per pair of sites: (25 + nseq) adds instead of 2 nseq.
# 8 waves x 80 sequences on 256 VGPRs: 160 accumulator + 50 table + 20 row registers.
LOGITS_CFG = {21: (8, 96), 5: (16, 48), 25: (8, 80)}
LOGITS_PAIR_Q = 5              # states per site of the pair variant
# Row registers double-buffered (round 4): the q rows of site j+1 are requested WHILE site j's adds issue (one ds_read every
# few adds, into a second set of q register pairs), so a wave no longer stops for an LDS round trip (~490 clk) in front of
# every site -- with two waves per SIMD that bubble is what kept the adds at ~57 % of the VALU.  The second set costs 2q
# registers, i.e. q = 21 runs 80 instead of 96 sequences per wave; only the first site of a tile still waits.
LOGITS_DBUF = {21: False, 5: False, 25: False}
if os.environ.get("DCA_GEN_LG21"):             # experiments: "waves,nseq,dbuf"
    _w, _n, _d = (int(v) for v in os.environ["DCA_GEN_LG21"].split(","))
    LOGITS_CFG[21] = (_w, _n)
    LOGITS_DBUF[21] = bool(_d)
LS0 = 36                       # first state-word SGPR
LOGITS_CHUNK = 16              # largest scalar load of the single-set schedule (16 dwords = 32 sequences)


def logits_single_set(q):
    return LOGITS_CFG[q][0] == 8


def logits_chunks(q):
    """single-set schedule: the site's nseq/2 state dwords as (first dword, dwords) pieces of 16 / 8 / 4, smallest first (the
    LAST piece should be a long one: the next site's first pieces are requested when its first word is used)"""
    nw = LOGITS_CFG[q][1] // 2
    sizes, left = [], nw
    for piece in (16, 8, 4):
        while left >= piece:
            sizes.append(piece)
            left -= piece
    assert left == 0 and len(sizes) >= 2
    sizes.sort()
    out, at = [], 0
    for sz in sizes:
        out.append((at, sz))
        at += sz
    return out


def logits_plan(q):
    """-> (waves, nseq, first row register, first accumulator register, words per SGPR set, first temp SGPR)"""
    waves, nseq = LOGITS_CFG[q]
    vg = {16: 128, 12: 168, 8: 256}[waves]
    acc0 = vg - 2 * nseq
    nw = (nseq // 2 + 3) // 4 * 4
    tb = LS0 + (1 if logits_single_set(q) else 2) * nw
    assert tb + 7 <= 96 and nseq % 8 == 0
    w0 = acc0 - 2 * q * (2 if LOGITS_DBUF[q] else 1)
    assert logits_temp0(q, w0) >= 8, "no VGPRs left for the compiler"
    return waves, nseq, w0, acc0, nw, tb


def logits_temp0(q, w0):
    """first register the block clobbers: the pair variant keeps the 2 x 5 rows it builds its table from below the table"""
    return w0 - (4 * LOGITS_PAIR_Q if q == 25 else 0)


def logits_chunk_load(q, c):
    """single-set schedule: scalar load of piece c of the site the state pointer is at"""
    waves, nseq, w0, acc0, nw, tb = logits_plan(q)
    at, sz = logits_chunks(q)[c]
    base = LS0 + at
    assert base % 4 == 0
    return "s_load_dwordx%d s[%d:%d], s[%d:%d], 0x%x" % (sz, base, base + sz - 1, tb + 2, tb + 3, at * 4)


def logits_state_loads(q, sset):
    """scalar loads of one site's nseq 16-bit M0 images (nseq/2 dwords) into SGPR set sset"""
    waves, nseq, w0, acc0, nw, tb = logits_plan(q)
    base = LS0 + sset * nw
    r, left, at = [], nseq // 2, 0
    for piece in (16, 8, 4, 2, 1):
        while left >= piece:
            reg = "s[%d:%d]" % (base + at, base + at + piece - 1) if piece > 1 else "s%d" % (base + at)
            r.append("s_load_dword%s %s, s[%d:%d], 0x%x" % ("x%d" % piece if piece > 1 else "", reg, tb + 2, tb + 3, at * 4))
            at += piece
            left -= piece
    return r


LOGITS_PIECES = 64             # 1 KiB LDS-DMA pieces per tile (128 rows x 512 bytes)


def logits_stage_sites(q):
    """sites of the tile at whose start a wave issues its i-th LDS-DMA piece of the NEXT tile"""
    waves = LOGITS_CFG[q][0]
    ppw = (LOGITS_PIECES + waves - 1) // waves
    jt = logits_jt(q)
    return [i * jt // ppw for i in range(ppw)]


def logits_body(q, f64):
    """One tile.  Besides the adds, the wave issues its share of the LDS-DMA pieces of the next tile, one
    piece at the start of a site (logits_stage_sites) instead of all of them at the start of the tile:
    a burst of 64 KiB of LDS writes right after the barrier holds up every wave's row reads at once.

    State words.  Two-set plan: the next site's words are requested into the other set right after this site's
    `s_waitcnt`, a whole site ahead.  Single-set plan (the words are consumed in order, chunk by chunk): after the
    wait at the start of a site -- chunks 0 .. m-2 of this site have landed -- the LAST chunk of this site is requested
    (its registers were in use until the end of the previous site); a second wait in front of the last chunk's first use
    finds nothing else outstanding, and behind it chunks 0 .. m-2 of the NEXT site are requested into their by then
    free registers.  Every request is at least a chunk's worth of adds (plus the row fetch) ahead of the wait that
    covers it."""
    waves, nseq, w0, acc0, nw, tb = logits_plan(q)
    jt = logits_jt(q)
    single = logits_single_set(q)
    dbuf = LOGITS_DBUF[q]
    chunks = logits_chunks(q) if single else []
    nchunks = len(chunks)
    last_first_seq = 2 * chunks[-1][0] if single else -1          # first sequence whose state word lies in the last piece
    add = "v_add_f64" if f64 else "v_pk_add_f32"
    gb, ld = tb + 4, tb + 6
    stage_at = logits_stage_sites(q)
    advance = ["s_add_u32 s%d, s%d, %%[stride]" % (tb + 2, tb + 2), "s_addc_u32 s%d, s%d, 0" % (tb + 3, tb + 3)]
    rowsets = [w0, w0 + 2 * q] if dbuf else [w0, w0]

    pq = LOGITS_PAIR_Q
    rt = logits_temp0(q, w0)

    def row_reads(jj, rb):
        if q == 25:          # the 5 + 5 rows of the pair's two sites (tile rows 10 jj ..), into the registers below the table
            return ["ds_read_b64 v[%d:%d], %%[vbase] offset:%d" % (rt + 2 * k, rt + 2 * k + 1, (jj * 2 * pq + k) * ROWBYTES) for k in range(2 * pq)]
        return ["ds_read_b64 v[%d:%d], %%[vbase] offset:%d" % (rb + 2 * b, rb + 2 * b + 1, (jj * q + b) * ROWBYTES) for b in range(q)]

    def table_build(rb):     # "row" 5 b1 + b2 of the pair = row b1 of its first site + row b2 of its second
        return ["%s v[%d:%d], v[%d:%d], v[%d:%d]" % (add, rb + 2 * (pq * b1 + b2), rb + 2 * (pq * b1 + b2) + 1, rt + 2 * b1, rt + 2 * b1 + 1,
                                                    rt + 2 * (pq + b2), rt + 2 * (pq + b2) + 1) for b1 in range(pq) for b2 in range(pq)]

    o = ["s_mov_b32 s%d, m0" % (tb + 1),
         "s_mov_b64 s[%d:%d], %%[sptr]" % (tb + 2, tb + 3),
         "s_mov_b64 s[%d:%d], %%[gbase]" % (gb, gb + 1),
         "s_mov_b32 s%d, %%[ldst]" % ld,
         "s_mov_b32 s%d, 0" % tb]
    o += [logits_chunk_load(q, c) for c in range(nchunks - 1)] if single else logits_state_loads(q, 0)
    # pair variant: the 10 rows are consumed by the table build, so the NEXT pair's rows are requested right behind it, into
    # the same registers, and arrive under this pair's adds -- only the first pair of a tile waits for its LDS round trip
    rows_ahead = q == 25
    if dbuf or rows_ahead:
        o += row_reads(0, rowsets[0])
    for jj in range(jt):
        cur = LS0 if single else LS0 + (jj % 2) * nw
        rb = rowsets[jj % 2]
        if jj + 1 < jt and not single:
            o += advance
        for i, at in enumerate(stage_at):
            if at == jj:
                o += ["s_cmp_gt_u32 %%[npc], %d" % i,
                      "s_cbranch_scc0 .Ldca_lg_skip%d_%%=" % i,
                      "s_mov_b32 m0, s%d" % ld,
                      "s_nop 0",
                      "global_load_lds_dwordx4 %%[voff], s[%d:%d]" % (gb, gb + 1) + LG_DMA_AUX,
                      "s_add_u32 s%d, s%d, %%[ginc]" % (gb, gb),
                      "s_addc_u32 s%d, s%d, 0" % (gb + 1, gb + 1),
                      "s_add_u32 s%d, s%d, %d" % (ld, ld, waves * 1024),
                      ".Ldca_lg_skip%d_%%=:" % i]
        if not dbuf and not rows_ahead:
            o += row_reads(jj, rb)
        o.append("s_waitcnt lgkmcnt(0)")          # this site's rows (and the state words requested during the previous site) have landed
        if single:
            o.append(logits_chunk_load(q, nchunks - 1))
        elif jj + 1 < jt:
            o += logits_state_loads(q, (jj + 1) % 2)
        if q == 25:
            o += table_build(rb)
            if jj + 1 < jt:
                o += row_reads(jj + 1, rb)
        o.append("s_set_gpr_idx_on s%d, 0x2" % tb)
        # double-buffered rows: the next site's row reads are dealt over this site's adds (DS instructions are not indexed)
        pending = row_reads(jj + 1, rowsets[(jj + 1) % 2]) if dbuf and jj + 1 < jt else []
        every = max(1, (nseq - 16) // max(1, len(pending)))
        for sq in range(nseq):
            if single and sq == last_first_seq:
                o.append("s_waitcnt lgkmcnt(0)")
                if jj + 1 < jt:
                    o += advance + [logits_chunk_load(q, c) for c in range(nchunks - 1)]
            if pending and sq % every == 0:
                o.append(pending.pop(0))
            w = cur + sq // 2
            if sq % 2 == 0:
                o.append("s_pack_ll_b32_b16 m0, s%d, 0" % w)
            else:
                o.append("s_lshr_b32 m0, s%d, 16" % w)
            a = acc0 + 2 * sq
            o.append("%s v[%d:%d], v[%d:%d], v[%d:%d]" % (add, a, a + 1, a, a + 1, rb, rb + 1))
        o += pending
        o.append("s_set_gpr_idx_off")
    o.append("s_mov_b32 m0, s%d" % (tb + 1))
    return o


def logits_macro(q, f64):
    """VBASE: LDS address of the tile + 8 * lane.  SPTR / STRIDE: the wave's state words of the tile's first
    site / bytes between sites.  Staging of the next tile: NPC pieces (0: none), piece i copies 1 KiB from
    GBASE + i * GINC + VOFF (per lane) to LDS address LDST + i * waves * 1024."""
    waves, nseq, w0, acc0, nw, tb = logits_plan(q)
    jsfx = "J%d" % PAIR_JT if q == 25 and PAIR_JT != 12 else ""
    lines = ["#define DCA_LOGITS_Q%d%s_%s(VBASE, SPTR, STRIDE, NPC, GBASE, GINC, VOFF, LDST, ACC) \\" % (q, jsfx, "F64" if f64 else "F32"),
             "    asm volatile( \\"]
    for ln in logits_body(q, f64):
        lines.append('        "%s\\n" \\' % ln)
    ops = ['"+{v[%d:%d]}"((ACC).p[%d])' % (acc0 + 16 * i, acc0 + 16 * i + 15, i) for i in range(nseq // 8)]
    lines.append("        : %s \\" % ", ".join(ops))
    lines.append('        : [vbase] "v"(VBASE), [sptr] "s"(SPTR), [stride] "s"(STRIDE), [npc] "s"(NPC), [gbase] "s"(GBASE), '
                 '[ginc] "s"(GINC), [voff] "v"(VOFF), [ldst] "s"(LDST) \\')
    t0 = logits_temp0(q, w0)
    clob = ['"memory"', '"scc"'] + ['"v%d"' % (t0 + i) for i in range(acc0 - t0)] + ['"s%d"' % i for i in range(LS0, tb + 7)]
    lines.append("        : %s)" % ", ".join(clob))
    return "\n".join(lines)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    lout = ["// GENERATED by tools/gen_plm_asm.py -- do not edit by hand.",
            "// Inline-asm inner blocks of plm_logits_kernel; see the generator for the register plan.", ""]
    for q in (21, 5, 25):
        lout.append("#define DCA_LOGITS_WAVES_Q%d %d" % (q, LOGITS_CFG[q][0]))
        lout.append("#define DCA_LOGITS_NSEQ_Q%d %d" % (q, LOGITS_CFG[q][1]))
        for f64 in ((0, 1) if q != 25 else (0,)):          # the pair alphabet is a float32 formulation (it re-associates the sums)
            lout.append(logits_macro(q, f64))
            lout.append("")
    # the pair variant again with 11 and 10 pairs per tile: the engine takes the one that pads ceil(L / 2) least
    # (L = 150: 75 pairs are 7 tiles of 12 = 84 pair steps, or 7 of 11 = 77)
    global PAIR_JT
    for PAIR_JT in (11, 10):
        lout.append(logits_macro(25, 0))
        lout.append("")
    PAIR_JT = 12
    lpath = os.path.join(here, "..", "pydca_amd", "csrc", "logits_gather_asm.inc")
    with open(lpath, "w") as fh:
        fh.write("\n".join(lout))
    print("wrote", os.path.normpath(lpath))
    out = ["// GENERATED by tools/gen_plm_asm.py -- do not edit by hand.",
           "// Inline-asm gather blocks of plm_scatter_kernel; see the generator for the register plan.", ""]
    for q in (21, 5):                  # two sites per wave, state words by scalar loads (the product path;
        # macro() / body() generate the earlier v_readlane form, also with more sites per wave -- kept for
        # tools/experiments, not emitted)
        for f64 in (0, 1):
            out.append(macro_smem(q, f64))
            out.append("")
        # float64 only: workgroups of 8 / 4 waves (16 / 8 sites) for the parity mode's single chain per sum on alignments with
        # few column strips, where the tile-range split that would otherwise fill the chip is not allowed
        for waves in (8, 4):
            out.append(macro_smem(q, 1, waves))
            out.append("")
    # q = 5 in float32 on the site-pair alphabet: the block with 25 "states" per unit, a unit being a pair of sites (round 5)
    out.append(macro_smem(25, 0))
    out.append("")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pydca_amd", "csrc", "scatter_gather_asm.inc")
    with open(path, "w") as fh:
        fh.write("\n".join(out))
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
