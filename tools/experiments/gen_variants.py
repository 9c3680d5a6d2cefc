#!/usr/bin/env python3
"""Variants of the gather block for gpridx_bench2.hip (micro-benchmark of what bounds the loop).
usage: gen_variants.py VARIANT > gather_variant.inc
  full      : as the product kernel (M0 write + indexed pk_add per (row, site))
  nom0      : indexed pk_add, M0 written once per quarter (all rows add to one state)
  noidx     : M0 writes kept, index mode off (pk_add always to state 0)
  add2      : two v_add_f32 instead of v_pk_add_f32 (indexed)
  nolds     : no ds_read in the loop (adds a stale register)
  depth16   : full, 16 reads in flight
"""
import sys
sys.path.insert(0, "../")
ROWS, ROWBYTES, JW = 128, 512, 2
S0, T0 = 40, 72


def body(variant):
    q = 21
    DEPTH = 16 if variant == "depth16" else 8
    a1 = 128 - 2 * q
    a0 = a1 - 2 * q
    d0 = a0 - 2 * DEPTH
    acc = [a0, a1]
    o = ["s_mov_b32 s%d, m0" % (T0 + 1)]

    def ds(r):
        k = r % DEPTH
        return "ds_read_b64 v[%d:%d], %%[vbase] offset:%d" % (d0 + 2 * k, d0 + 2 * k + 1, r * ROWBYTES)

    for r in range(DEPTH):
        o.append(ds(r))
    quarter = ROWS // 4
    for r in range(ROWS):
        if r % quarter == 0:
            if r:
                o.append("s_set_gpr_idx_off")
            for jj in range(JW):
                for w in range(16):
                    o.append("v_readlane_b32 s%d, %%[st%d], %d" % (S0 + jj * 16 + w, jj, (r // quarter) * 16 + w))
            o.append("s_nop 3")
            o.append("s_mov_b32 s%d, 0" % T0)
            if variant != "noidx":
                o.append("s_set_gpr_idx_on s%d, 0x9" % T0)
        if variant == "wait4":
            if r % 4 == 0:
                o.append("s_waitcnt lgkmcnt(%d)" % max(0, min(DEPTH - 4, ROWS - 4 - r)))
        elif variant not in ("nolds", "nowait"):
            o.append("s_waitcnt lgkmcnt(%d)" % min(DEPTH - 1, ROWS - 1 - r))
        k = r % DEPTH
        for jj in range(JW):
            w = S0 + jj * 16 + (r % quarter) // 2
            if variant != "nom0":
                if variant == "noidx":
                    o.append("s_lshr_b32 s%d, s%d, 16" % (T0, w))
                elif r % 2 == 0:
                    o.append("s_pack_ll_b32_b16 m0, s%d, 0" % w)
                else:
                    o.append("s_lshr_b32 m0, s%d, 16" % w)
            if variant == "add2":
                o.append("v_add_f32 v%d, v%d, v%d" % (acc[jj], acc[jj], d0 + 2 * k))
                o.append("v_add_f32 v%d, v%d, v%d" % (acc[jj] + 1, acc[jj] + 1, d0 + 2 * k + 1))
            else:
                o.append("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (acc[jj], acc[jj] + 1, acc[jj], acc[jj] + 1, d0 + 2 * k, d0 + 2 * k + 1))
        if variant == "wait4":
            if r % 4 == 3:
                for rr in range(r - 3 + DEPTH, r + 1 + DEPTH):
                    if rr < ROWS:
                        o.append(ds(rr))
        elif r + DEPTH < ROWS and variant != "nolds":
            o.append(ds(r + DEPTH))
    if variant != "noidx":
        o.append("s_set_gpr_idx_off")
    o.append("s_mov_b32 m0, s%d" % (T0 + 1))
    clob = ['"memory"'] + ['"v%d"' % (d0 + i) for i in range(2 * DEPTH)] + ['"s%d"' % (S0 + i) for i in range(32)] + ['"s72"', '"s73"']
    return o, clob


def body_smem():
    """state words through scalar loads: set A = s[36:67], set B = s[68:99] (16 words x 2 sites each);
    quarter k uses set k%2; the loads for quarter k+1 are issued at the start of quarter k; the quarter
    boundary drains lgkmcnt(0) (LDS pipeline bubble) to make sure they have landed."""
    q, DEPTH = 21, 8
    a1 = 128 - 2 * q
    a0 = a1 - 2 * q
    d0 = a0 - 2 * DEPTH
    acc = [a0, a1]
    SA, SB, TZ, TM = 32, 64, 96, 97
    sets = (SA, SB)
    o = ["s_mov_b32 s%d, m0" % TM]

    def ds(r):
        k = r % DEPTH
        return "ds_read_b64 v[%d:%d], %%[vbase] offset:%d" % (d0 + 2 * k, d0 + 2 * k + 1, r * ROWBYTES)

    def sload(qk, base):
        return ["s_load_dwordx16 s[%d:%d], %%[sp0], 0x%x" % (base, base + 15, qk * 64),
                "s_load_dwordx16 s[%d:%d], %%[sp1], 0x%x" % (base + 16, base + 31, qk * 64)]

    o += sload(0, sets[0])
    o.append("s_waitcnt lgkmcnt(0)")
    for r in range(DEPTH):
        o.append(ds(r))
    quarter = ROWS // 4
    for r in range(ROWS):
        if r % quarter == 0:
            qk = r // quarter
            if r:
                o.append("s_set_gpr_idx_off")
                o.append("s_waitcnt lgkmcnt(0)")          # words of this quarter (and the 8 reads in flight)
            if qk + 1 < 4:
                o += sload(qk + 1, sets[(qk + 1) % 2])
            o.append("s_mov_b32 s%d, 0" % TZ)
            o.append("s_set_gpr_idx_on s%d, 0x9" % TZ)
        # with a scalar load possibly in flight the counter over-counts by up to 2: still safe (see DESIGN notes)
        o.append("s_waitcnt lgkmcnt(%d)" % min(DEPTH - 1, ROWS - 1 - r))
        k = r % DEPTH
        cur = sets[(r // quarter) % 2]
        for jj in range(JW):
            w = cur + jj * 16 + (r % quarter) // 2
            o.append(("s_pack_ll_b32_b16 m0, s%d, 0" if r % 2 == 0 else "s_lshr_b32 m0, s%d, 16") % w)
            o.append("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (acc[jj], acc[jj] + 1, acc[jj], acc[jj] + 1, d0 + 2 * k, d0 + 2 * k + 1))
        if r + DEPTH < ROWS:
            o.append(ds(r + DEPTH))
    o.append("s_set_gpr_idx_off")
    o.append("s_mov_b32 m0, s%d" % TM)
    clob = ['"memory"'] + ['"v%d"' % (d0 + i) for i in range(2 * DEPTH)] + ['"s%d"' % i for i in range(32, 98)]
    return o, clob


if sys.argv[1] == "smem":
    o, clob = body_smem()
    print("#define GATHER_BLOCK_SMEM(VBASE, SP0, SP1, A0, B0, C0, A1, B1, C1) asm volatile( \\")
    for ln in o:
        print('    "%s\\n" \\' % ln)
    print('    : "+{v[44:75]}"(A0), "+{v[76:83]}"(B0), "+{v[84:85]}"(C0), "+{v[86:117]}"(A1), "+{v[118:125]}"(B1), "+{v[126:127]}"(C1) \\')
    print('    : [vbase] "v"(VBASE), [sp0] "s"(SP0), [sp1] "s"(SP1) \\')
    print("    : %s)" % ", ".join(clob))
    sys.exit(0)
o, clob = body(sys.argv[1])
print("#define GATHER_BLOCK(VBASE, ST0, ST1, A0, B0, C0, A1, B1, C1) asm volatile( \\")
for ln in o:
    print('    "%s\\n" \\' % ln)
print('    : "+{v[44:75]}"(A0), "+{v[76:83]}"(B0), "+{v[84:85]}"(C0), "+{v[86:117]}"(A1), "+{v[118:125]}"(B1), "+{v[126:127]}"(C1) \\')
print('    : [vbase] "v"(VBASE), [st0] "v"(ST0), [st1] "v"(ST1) \\')
print("    : %s)" % ", ".join(clob))
