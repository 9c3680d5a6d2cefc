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
S0, T0 = 40, 72


def tuples(q):
    """legal VGPR tuple sizes covering 2q registers"""
    return {21: [32, 8, 2], 5: [8, 2]}[q]


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


def body_smem(q, f64):
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

    o += sload(0, sets[0])
    for r in range(DEPTH):
        o.append(ds(r))
    quarter = ROWS // 4
    for r in range(ROWS):
        if r % quarter == 0:
            qk = r // quarter
            if r:
                o.append("s_set_gpr_idx_off")
            o.append("s_waitcnt lgkmcnt(0)")
            if qk + 1 < 4:
                o += sload(qk + 1, sets[(qk + 1) % 2])
            o.append("s_set_gpr_idx_on s%d, 0x9" % sets[qk % 2])       # index bits are rewritten before every use
        o.append("s_waitcnt lgkmcnt(%d)" % min(DEPTH - 1, ROWS - 1 - r))
        k = r % DEPTH
        cur = sets[(r // quarter) % 2]
        for jj in range(2):
            w = cur + jj * 16 + (r % quarter) // 2
            if r % 2 == 0:
                o.append("s_pack_ll_b32_b16 m0, s%d, 0" % w)
            else:
                o.append("s_lshr_b32 m0, s%d, 16" % w)
            o.append("%s v[%d:%d], v[%d:%d], v[%d:%d]" % (add, acc[jj], acc[jj] + 1, acc[jj], acc[jj] + 1,
                                                         d0 + 2 * k, d0 + 2 * k + 1))
        if r + DEPTH < ROWS:
            o.append(ds(r + DEPTH))
    o.append("s_set_gpr_idx_off")
    o.append("s_mov_b32 m0, vcc_lo")
    return o


def macro_smem(q, f64):
    d0, acc = plan(q, 2)
    names = "ABC"[:len(tuples(q))]
    params = ["VBASE", "SP0", "SP1"] + ["%s%d" % (n, jj) for jj in range(2) for n in names]
    lines = ["#define DCA_GATHER_Q%d_%s_SMEM(%s) \\" % (q, "F64" if f64 else "F32", ", ".join(params)), "    asm volatile( \\"]
    for ln in body_smem(q, f64):
        lines.append('        "%s\\n" \\' % ln)
    outs = []
    for jj, base in enumerate(acc):
        r = base
        for n, sz in zip(names, tuples(q)):
            outs.append('"+{v[%d:%d]}"(%s%d)' % (r, r + sz - 1, n, jj))
            r += sz
    lines.append("        : %s \\" % ", ".join(outs))
    lines.append('        : [vbase] "v"(VBASE), [sp0] "s"(SP0), [sp1] "s"(SP1) \\')
    clob = ['"memory"', '"vcc"'] + ['"v%d"' % (d0 + i) for i in range(2 * DEPTH)] + ['"s%d"' % i for i in range(SET_A, SET_B + 32)]
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
# One block = one wave, one LDS tile of JT sites, NB = 32 sequences, one 512-byte column strip.
# For every site j of the tile:
#
#     w[b] = Wtile[(j, b)][lane]  for b in 0..q-1      (q ds_read_b64 with immediate offsets)
#     for s in 0..31:  acc[s] += w[state(n0 + s, j)]   (source register selected through M0)
#
# i.e. the transpose of the scatter block: there the destination is indexed, here the source
# (src1 relative: M0 image 0x2000 | 2 x state).  The q rows of W of the site sit in q register
# pairs, so a (sequence, site) pair costs one SALU write of M0 and one packed add and no LDS access
# of its own; the LDS reads are q per 32 pairs.  The 32 M0 images of a site arrive with one
# s_load_dwordx16 that is issued one site ahead (two SGPR sets, ping-pong), so only the LDS latency
# of the q row reads is exposed per site -- the other three waves of the SIMD cover it.
#
# Register plan: accumulators v[64:127] (two 32-register tuples), w rows v[64-2q : 63],
# state words s[40:55] / s[56:71], temporaries s72 (zero), s73 (saved M0), s[74:75] (state pointer).
NBSEQ = 32


def logits_jt(q):
    return {21: 6, 5: 25}[q]


def logits_body(q, f64):
    w0 = 64 - 2 * q
    jt = logits_jt(q)
    add = "v_add_f64" if f64 else "v_pk_add_f32"
    sets = (S0, S0 + 16)
    o = ["s_mov_b32 s%d, m0" % (T0 + 1),
         "s_mov_b64 s[74:75], %[sptr]",
         "s_mov_b32 s%d, 0" % T0,
         "s_load_dwordx16 s[%d:%d], s[74:75], 0x0" % (sets[0], sets[0] + 15)]
    for jj in range(jt):
        cur = sets[jj % 2]
        nxt = sets[(jj + 1) % 2]
        if jj + 1 < jt:
            o.append("s_add_u32 s74, s74, %[stride]")
            o.append("s_addc_u32 s75, s75, 0")
        for b in range(q):
            o.append("ds_read_b64 v[%d:%d], %%[vbase] offset:%d" % (w0 + 2 * b, w0 + 2 * b + 1, (jj * q + b) * ROWBYTES))
        o.append("s_waitcnt lgkmcnt(0)")
        if jj + 1 < jt:
            o.append("s_load_dwordx16 s[%d:%d], s[74:75], 0x0" % (nxt, nxt + 15))
        o.append("s_set_gpr_idx_on s%d, 0x2" % T0)
        for sq in range(NBSEQ):
            w = cur + sq // 2
            if sq % 2 == 0:
                o.append("s_pack_ll_b32_b16 m0, s%d, 0" % w)
            else:
                o.append("s_lshr_b32 m0, s%d, 16" % w)
            a = 64 + 2 * sq
            o.append("%s v[%d:%d], v[%d:%d], v[%d:%d]" % (add, a, a + 1, a, a + 1, w0, w0 + 1))
        o.append("s_set_gpr_idx_off")
    o.append("s_mov_b32 m0, s%d" % (T0 + 1))
    return o


def logits_macro(q, f64):
    w0 = 64 - 2 * q
    lines = ["#define DCA_LOGITS_Q%d_%s(VBASE, SPTR, STRIDE, ACCA, ACCB) \\" % (q, "F64" if f64 else "F32"), "    asm volatile( \\"]
    for ln in logits_body(q, f64):
        lines.append('        "%s\\n" \\' % ln)
    lines.append('        : "+{v[64:95]}"(ACCA), "+{v[96:127]}"(ACCB) \\')
    lines.append('        : [vbase] "v"(VBASE), [sptr] "s"(SPTR), [stride] "s"(STRIDE) \\')
    clob = ['"memory"'] + ['"v%d"' % (w0 + i) for i in range(2 * q)] + ['"s%d"' % (S0 + i) for i in range(32)] + \
           ['"s%d"' % (T0 + i) for i in range(4)]
    lines.append("        : %s)" % ", ".join(clob))
    return "\n".join(lines)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    lout = ["// GENERATED by tools/gen_plm_asm.py -- do not edit by hand.",
            "// Inline-asm inner blocks of plm_logits_kernel; see the generator for the register plan.", ""]
    for q in (21, 5):
        for f64 in (0, 1):
            lout.append(logits_macro(q, f64))
            lout.append("")
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
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pydca_amd", "csrc", "scatter_gather_asm.inc")
    with open(path, "w") as fh:
        fh.write("\n".join(out))
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
