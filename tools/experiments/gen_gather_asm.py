#!/usr/bin/env python3
"""Generator of the inline-asm gather block used by the scatter kernel (and by the
micro-benchmark gpridx_bench.hip).  One block = one wave, one 128-row LDS tile, JW sites:

    for r in 0..127:  acc[site][state(site, r)] += tile[r][lane]        (8 bytes per lane)

The accumulator of a state is selected with the gfx9 VGPR index mode (s_set_gpr_idx_*), so the
rows are visited in file order: no sorted lists, no per-state loops, the LDS address of row r is
an immediate offset.  Per row: 1 ds_read_b64 + JW x (s_bfe, s_set_gpr_idx_idx, v_pk_add_f32).

Register plan (physical, fixed):  data ring v[D0 .. D0+2*DEPTH), accumulators of site jj at
v[A0 + jj*2*Q ...], 16 state words per half in s[S0 .. S0+16*JW), temporaries s[T0], s[T0+1].
"""
import sys


def gen(Q, JW, f64, D0, A0, S0, T0, depth=8, rows=128, rowbytes=512):
    o = []
    add = "v_add_f64" if f64 else "v_pk_add_f32"
    o.append("s_mov_b32 s%d, m0" % (T0 + 1))
    issued = 0

    def ds(r):
        k = r % depth
        return "ds_read_b64 v[%d:%d], %%[vbase] offset:%d" % (D0 + 2 * k, D0 + 2 * k + 1, r * rowbytes)

    for r in range(min(depth, rows)):
        o.append(ds(r))
        issued += 1
    half = rows // 2
    for r in range(rows):
        if r % half == 0:
            if r:
                o.append("s_set_gpr_idx_off")
            for jj in range(JW):
                for w in range(16):
                    o.append("v_readlane_b32 s%d, %%[st%d], %d" % (S0 + jj * 16 + w, jj, (r // half) * 16 + w))
            o.append("s_nop 3")
            o.append("s_set_gpr_idx_on s%d, 0x9" % T0 if False else "s_mov_b32 s%d, 0" % T0)
            o.append("s_set_gpr_idx_on s%d, 0x9" % T0)
        inflight_after = min(depth - 1, rows - 1 - r)
        o.append("s_waitcnt lgkmcnt(%d)" % inflight_after)
        k = r % depth
        for jj in range(JW):
            w = (r % half) // 4
            o.append("s_bfe_u32 s%d, s%d, 0x%x" % (T0, S0 + jj * 16 + w, ((r % 4) * 8) | (8 << 16)))
            o.append("s_set_gpr_idx_idx s%d" % T0)
            a = A0 + jj * 2 * Q
            o.append("%s v[%d:%d], v[%d:%d], v[%d:%d]" % (add, a, a + 1, a, a + 1, D0 + 2 * k, D0 + 2 * k + 1))
        if r + depth < rows:
            o.append(ds(r + depth))
    o.append("s_set_gpr_idx_off")
    o.append("s_mov_b32 m0, s%d" % (T0 + 1))
    return o


def clobbers(Q, JW, D0, A0, S0, T0, depth=8):
    c = ["v%d" % (D0 + i) for i in range(2 * depth)]
    c += ["s%d" % (S0 + i) for i in range(16 * JW)] + ["s%d" % T0, "s%d" % (T0 + 1)]
    return c


if __name__ == "__main__":
    Q, JW, f64, D0, A0, S0, T0 = (int(x) for x in sys.argv[1:8])
    lines = gen(Q, JW, f64, D0, A0, S0, T0)
    print("\n".join('"%s\\n"' % ln for ln in lines))
