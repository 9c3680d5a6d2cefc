/* TEST INFRASTRUCTURE ONLY -- CPU oracle, not part of the shipped product.
 *
 * Precision-generic body of the plmDCA oracle.  Included twice by plm_oracle.c
 * with REAL=float (reference precision) and REAL=double (deterministic parity
 * target).  Every function cites the reference lines it
 * restates; paths are relative to /root/reference/pydca/plmdca/.
 *
 * Nothing here is copied from the reference: it is a restatement of the
 * algorithm in plain C with its own data layout (uint8 MSA, flat arrays).
 */
#ifndef REAL
#error "include from plm_oracle.c"
#endif

/* (s, c) += v : plain addition in the reference's order, or Neumaier's compensated sum (see plm_oracle.c) */
#undef FX_ADD
#ifdef ORACLE_COMPENSATED_FX
#define FX_ADD(s, c, v) do { const REAL _v = (v); const REAL _t = (s) + _v; \
        if (REAL_ABS(s) >= REAL_ABS(_v)) (c) += ((s) - _t) + _v; else (c) += (_v - _t) + (s); (s) = _t; } while (0)
#else
#define FX_ADD(s, c, v) do { (s) += (v); } while (0)
#endif

/* ---- sequence weights: plmdca_numerics.cpp:611-671 (OpenMP branch :627-645)
 * and meanfield_dca/msa_numerics.py:13-50 for the double variant.
 * count_n = #{m : (REAL)ident(n,m)/(REAL)L > seqid}, self included; w = 1/count. */
void FN(oracle_weights)(const uint8_t* X, int N, int L, REAL seqid, REAL* w, int threads)
{
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int n = 0; n < N; ++n) {
        const uint8_t* a = X + (size_t)n * L;
        REAL cnt = 0;
        for (int m = 0; m < N; ++m) {
            const uint8_t* b = X + (size_t)m * L;
            unsigned ident = 0;
            for (int s = 0; s < L; ++s) ident += (a[s] == b[s]);
            REAL sim = (REAL)ident / (REAL)L;
            if (sim > seqid) cnt += 1;
        }
        w[n] = (REAL)1 / cnt;
    }
}

/* ---- Meff: plmdca_numerics.cpp:46 (sequential sum in REAL) */
REAL FN(oracle_meff)(const REAL* w, int N)
{
    REAL s = 0;
    for (int n = 0; n < N; ++n) s += w[n];
    return s;
}

/* ---- initial parameters: getSingleSiteFreqs :51-81 + initFieldsAndCouplings :207-249
 * h_i(a) = log(f_i(a)*Meff + 1) - mean_a(...);  J = 0. */
void FN(oracle_init_x)(const uint8_t* X, const REAL* w, int N, int L, int q, REAL* x)
{
    const size_t P = plm_num_params(L, q);
    REAL meff = FN(oracle_meff)(w, N);
    for (int i = 0; i < L; ++i) {
        REAL* h = x + (size_t)i * q;
        for (int a = 0; a < q; ++a) h[a] = 0;
        for (int n = 0; n < N; ++n) h[X[(size_t)n * L + i]] += w[n];
        for (int a = 0; a < q; ++a) h[a] /= meff;
        for (int a = 0; a < q; ++a) h[a] = REAL_LOG(h[a] * meff + (REAL)1);
    }
    for (int i = 0; i < L; ++i) {
        REAL* h = x + (size_t)i * q;
        REAL s = 0;
        for (int a = 0; a < q; ++a) s += h[a];
        REAL av = s / (REAL)q;
        for (int a = 0; a < q; ++a) h[a] -= av;
    }
    for (size_t k = (size_t)L * q; k < P; ++k) x[k] = 0;
}

/* ---- objective + "gradient": PlmDCA::gradient, plmdca_numerics.cpp:436-607.
 * carry != 0 reproduces the reference's un-reset prob_ni buffer (:492,:499):
 * the logits of sequence n at site i start from the normalised probabilities
 * of sequence n-1 at the same site.  carry == 0 is the mathematically exact
 * pseudolikelihood gradient (opt-in mode of the product).
 * Site-parallel like the reference (:490); the per-site buffers are merged in
 * ascending site order afterwards, i.e. the reference's 1-thread order. */
__attribute__((target_clones("avx2", "default")))      /* same IEEE operations, wider registers where the host has them */
REAL FN(oracle_gradient)(const uint8_t* X, const REAL* w, int N, int L, int q,
                         REAL lambda_h, REAL lambda_J, const REAL* x, REAL* g,
                         int carry, int threads)
{
    const size_t nh = (size_t)L * q;
    const size_t q2 = (size_t)q * q;
    const size_t npairs = (size_t)L * (L - 1) / 2;
    REAL fx = 0, fxc = 0;      /* fxc: compensation term (stays 0 in the reference-order float instantiation) */

    /* L2 terms, :463-486 -- sequential REAL sums in parameter order */
    for (size_t k = 0; k < nh; ++k) {
        g[k] = (REAL)2 * lambda_h * x[k];
        FX_ADD(fx, fxc, lambda_h * x[k] * x[k]);
    }
    for (size_t k = nh; k < nh + npairs * q2; ++k) {
        g[k] = (REAL)2 * lambda_J * x[k];
        FX_ADD(fx, fxc, lambda_J * x[k] * x[k]);
    }

    /* per-site scratch: cg[i] holds L*q*q entries in pair orientation
     * (state of the smaller site first), as :494,:541-567 */
    REAL* cg = (REAL*)malloc((size_t)L * L * q2 * sizeof(REAL));     /* every block a site owns is written below; [i][i] is never read */
    REAL* hg = (REAL*)calloc(nh, sizeof(REAL));
    REAL* fsite = (REAL*)calloc(2 * (size_t)L, sizeof(REAL));      /* per site: sum, compensation */
    if (!cg || !hg || !fsite) { free(cg); free(hg); free(fsite); return (REAL)NAN; }

    /* Per site the reference walks x with stride q for the partners j > i (:515, :553-563).  Here every thread first copies
     * its site's L coupling blocks into a site-local table Wi[j][b][a] = J(i:a, j:b) whose rows are contiguous in a, and
     * accumulates into a table Ti of the same shape that is written back into the pair orientation afterwards.  Every
     * scalar still receives the same additions in the same order as in the reference (partners in ascending j, the
     * "-= w" of :541-551 before the "+= w p" of :553-566 within a sequence), so the results are bit-identical to the
     * strided form -- test_oracle_golden pins that against the reference's own float32 runs -- at a third of the time. */
    int failed = 0;
#pragma omp parallel num_threads(threads)
    {
        REAL* Wi = (REAL*)malloc((size_t)L * q2 * sizeof(REAL));
        REAL* Ti = (REAL*)malloc((size_t)L * q2 * sizeof(REAL));
#ifdef ORACLE_CANONICAL_F64
        REAL* Tt = (REAL*)malloc((size_t)L * q2 * sizeof(REAL));      /* running sum of the finished blocks */
        if (!Tt) { free(Ti); Ti = NULL; }
#endif
        if (!Wi || !Ti) {
#pragma omp atomic write
            failed = 1;
        }
#pragma omp barrier
#pragma omp for schedule(dynamic, 1)
        for (int i = 0; i < L; ++i) {
            if (failed) continue;
            REAL p[64];
            REAL* cgi = cg + (size_t)i * L * q2;
            REAL* hgi = hg + (size_t)i * q;
            REAL fi = 0, fic = 0;
#ifdef ORACLE_CANONICAL_F64
            REAL hgc[64];                         /* compensation terms of the field-gradient sums */
            for (int a = 0; a < q; ++a) hgc[a] = 0;
#endif
            for (int j = 0; j < i; ++j) memcpy(Wi + (size_t)j * q2, x + nh + plm_pair_index(L, j, i) * q2, q2 * sizeof(REAL));
            for (int j = i + 1; j < L; ++j) {
                const REAL* Jij = x + nh + plm_pair_index(L, i, j) * q2;
                REAL* Wj = Wi + (size_t)j * q2;
                for (int a = 0; a < q; ++a) for (int b = 0; b < q; ++b) Wj[(size_t)b * q + a] = Jij[(size_t)a * q + b];
            }
            memset(Ti, 0, (size_t)L * q2 * sizeof(REAL));
#ifdef ORACLE_CANONICAL_F64
            int blocksDone = 0;                   /* blocks of ORACLE_CANONICAL_BLOCK sequences already folded into Tt */
#endif
            for (int a = 0; a < q; ++a) p[a] = 0;
            for (int n = 0; n < N; ++n) {
                const uint8_t* s = X + (size_t)n * L;
                if (!carry) for (int a = 0; a < q; ++a) p[a] = 0;
#ifdef ORACLE_CANONICAL_F64
                /* float64 instantiation: the same terms in the order ((sum_j J) + h) + carry -- the coupling rows first, in
                 * ascending j from zero, then the field, then the carried-over probabilities (see plm_oracle.c) */
                REAL sj[64];
                for (int a = 0; a < q; ++a) sj[a] = 0;
                for (int j = 0; j < i; ++j) {
                    const REAL* r = Wi + (size_t)j * q2 + (size_t)s[j] * q;
                    for (int a = 0; a < q; ++a) sj[a] += r[a];
                }
                for (int j = i + 1; j < L; ++j) {
                    const REAL* r = Wi + (size_t)j * q2 + (size_t)s[j] * q;
                    for (int a = 0; a < q; ++a) sj[a] += r[a];
                }
                for (int a = 0; a < q; ++a) p[a] = (sj[a] + x[(size_t)i * q + a]) + p[a];
#else
                for (int a = 0; a < q; ++a) p[a] += x[(size_t)i * q + a];
                for (int j = 0; j < i; ++j) {
                    const REAL* r = Wi + (size_t)j * q2 + (size_t)s[j] * q;
                    for (int a = 0; a < q; ++a) p[a] += r[a];
                }
                for (int j = i + 1; j < L; ++j) {
                    const REAL* r = Wi + (size_t)j * q2 + (size_t)s[j] * q;
                    for (int a = 0; a < q; ++a) p[a] += r[a];
                }
#endif
                REAL mx = p[0];
                for (int a = 0; a < q; ++a) if (p[a] > mx) mx = p[a];
                for (int a = 0; a < q; ++a) p[a] = REAL_EXP(p[a] - mx);
                REAL z = 0;
                for (int a = 0; a < q; ++a) z += p[a];
                z = (REAL)1 / z;
                for (int a = 0; a < q; ++a) p[a] *= z;

                const REAL wn = w[n];
                const int ri = s[i];
                FX_ADD(fi, fic, -(wn * REAL_LOG(p[ri])));
                REAL wp[64];
#ifdef ORACLE_CANONICAL_F64
                /* float64 instantiation: every sum receives the ONE rounded residual w p(a) - w delta(a, x_ni) instead of the
                 * two addends -w and +w p(a) of :541-566 one after the other; the field-gradient sums are compensated */
                for (int a = 0; a < q; ++a) wp[a] = wn * p[a];
                wp[ri] -= wn;
                for (int a = 0; a < q; ++a) FX_ADD(hgi[a], hgc[a], wp[a]);
                for (int j = 0; j < L; ++j) {
                    if (j == i) continue;
                    REAL* row = Ti + (size_t)j * q2 + (size_t)s[j] * q;
                    for (int a = 0; a < q; ++a) row[a] += wp[a];
                }
#else
                hgi[ri] -= wn;
                for (int a = 0; a < q; ++a) hgi[a] += wn * p[a];
                for (int a = 0; a < q; ++a) wp[a] = wn * p[a];
                for (int j = 0; j < L; ++j) {
                    if (j == i) continue;
                    REAL* row = Ti + (size_t)j * q2 + (size_t)s[j] * q;
                    row[ri] -= wn;
                    for (int a = 0; a < q; ++a) row[a] += wp[a];
                }
#endif
#ifdef ORACLE_CANONICAL_F64
                /* float64 instantiation: the per-slot chains run over BLOCKS of ORACLE_CANONICAL_BLOCK consecutive sequences,
                 * each summed from zero in ascending n, and the block sums are added in ascending block order
                 * (((B0 + B1) + B2) + ...) -- see plm_oracle.c.  An alignment of at most one block is one plain chain. */
                if ((n + 1) % ORACLE_CANONICAL_BLOCK == 0 && n + 1 < N) {
                    if (blocksDone == 0) memcpy(Tt, Ti, (size_t)L * q2 * sizeof(REAL));
                    else for (size_t k = 0; k < (size_t)L * q2; ++k) Tt[k] += Ti[k];
                    memset(Ti, 0, (size_t)L * q2 * sizeof(REAL));
                    ++blocksDone;
                }
#endif
            }
#ifdef ORACLE_CANONICAL_F64
            if (blocksDone > 0) for (size_t k = 0; k < (size_t)L * q2; ++k) Ti[k] = Tt[k] + Ti[k];     /* the last block */
#endif
            /* back into the pair orientation (state of the smaller site first), as :494,:541-567 */
            for (int j = 0; j < i; ++j) memcpy(cgi + (size_t)j * q2, Ti + (size_t)j * q2, q2 * sizeof(REAL));
            for (int j = i + 1; j < L; ++j) {
                const REAL* Tj = Ti + (size_t)j * q2;
                REAL* c = cgi + (size_t)j * q2;
                for (int a = 0; a < q; ++a) for (int b = 0; b < q; ++b) c[(size_t)a * q + b] = Tj[(size_t)b * q + a];
            }
            fsite[2 * i] = fi;
            fsite[2 * i + 1] = fic;
#ifdef ORACLE_CANONICAL_F64
            for (int a = 0; a < q; ++a) hgi[a] += hgc[a];
#endif
        }
        free(Wi); free(Ti);
#ifdef ORACLE_CANONICAL_F64
        free(Tt);
#endif
    }
    if (failed) { free(cg); free(hg); free(fsite); return (REAL)NAN; }

    /* merge, :570-602, in ascending site order (deterministic) */
    for (int i = 0; i < L; ++i) {
        FX_ADD(fx, fxc, fsite[2 * i]);
        fxc += fsite[2 * i + 1];
        for (int a = 0; a < q; ++a) g[(size_t)i * q + a] += hg[(size_t)i * q + a];
    }
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int i = 0; i < L; ++i) {
        for (int j = i + 1; j < L; ++j) {
            REAL* gp = g + nh + plm_pair_index(L, i, j) * q2;
            const REAL* ci = cg + ((size_t)i * L + j) * q2; /* site i's view of (i,j) */
            const REAL* cj = cg + ((size_t)j * L + i) * q2; /* site j's view of (i,j) */
            for (size_t k = 0; k < q2; ++k) { gp[k] += ci[k]; gp[k] += cj[k]; }
        }
    }
    free(cg); free(hg); free(fsite);
    return fx + fxc;
}

/* ======================================================================
 * L-BFGS with More-Thuente line search.
 * Restates what the reference backend runs: lbfgs/lib/lbfgs.cpp:248-644 (driver),
 * :815-1004 (line search), :1128-1295 (trial interval update), with the
 * parameters set in plmdcaBackend.cpp:68-75 and the defaults of lbfgs.cpp:116-121.
 * Vector reductions are sequential sums in REAL (arithmetic_ansi.h:114-121); the float64 instantiation compensates
 * them (Neumaier, like the objective: see plm_oracle.c), so that they do not depend on the order of summation.
 * ====================================================================== */

static REAL FN(vdot)(const REAL* a, const REAL* b, size_t n)
{
    REAL s = 0, c = 0;      /* c: compensation term (stays 0 in the reference-order float instantiation) */
    for (size_t i = 0; i < n; ++i) FX_ADD(s, c, a[i] * b[i]);     /* the product is rounded first (-ffp-contract=off) */
    return s + c;
}

/* minimiser of the cubic through (u,fu,du),(v,fv,dv): lbfgs.cpp:1024-1038 */
static REAL FN(cubic_min)(REAL u, REAL fu, REAL du, REAL v, REAL fv, REAL dv)
{
    REAL d = v - u;
    REAL theta = (fu - fv) * 3 / d + du + dv;
    REAL p = REAL_ABS(theta), qq = REAL_ABS(du), r = REAL_ABS(dv);
    REAL s = p >= qq ? p : qq; s = s >= r ? s : r;
    REAL a = theta / s;
    REAL gamma = s * REAL_SQRT(a * a - (du / s) * (dv / s));
    if (v < u) gamma = -gamma;
    p = gamma - du + theta;
    qq = gamma - du + gamma + dv;
    r = p / qq;
    return u + r * d;
}

/* safeguarded cubic for the "derivative shrinks" case: lbfgs.cpp:1052-1072 */
static REAL FN(cubic_min_clamped)(REAL u, REAL fu, REAL du, REAL v, REAL fv, REAL dv,
                                  REAL lo, REAL hi)
{
    REAL d = v - u;
    REAL theta = (fu - fv) * 3 / d + du + dv;
    REAL p = REAL_ABS(theta), qq = REAL_ABS(du), r = REAL_ABS(dv);
    REAL s = p >= qq ? p : qq; s = s >= r ? s : r;
    REAL a = theta / s;
    REAL rad = a * a - (du / s) * (dv / s);
    REAL gamma = s * REAL_SQRT(rad > 0 ? rad : 0);
    if (u < v) gamma = -gamma;
    p = gamma - dv + theta;
    qq = gamma - dv + gamma + du;
    r = p / qq;
    if (r < 0. && gamma != 0.) return v - r * d;
    if (a < 0) return hi;
    return lo;
}

/* quadratic through (u,fu,du),(v,fv): lbfgs.cpp:1083-1085 */
static REAL FN(quad_min_f)(REAL u, REAL fu, REAL du, REAL v, REAL fv)
{
    REAL a = v - u;
    return u + du / ((fu - fv) / a + du) / 2 * a;
}

/* secant through the two derivatives: lbfgs.cpp:1095-1097 */
static REAL FN(quad_min_d)(REAL u, REAL du, REAL v, REAL dv)
{
    REAL a = u - v;
    return v + dv / (dv - du) * a;
}

typedef struct {
    REAL st, f, d;
} FN(lspoint);

/* More-Thuente trial-interval update: lbfgs.cpp:1128-1295.
 * best/other are the interval end points, *t/ft/dt the current trial. */
static int FN(mt_update)(FN(lspoint)* best, FN(lspoint)* other, REAL* t, REAL ft, REAL dt,
                         REAL tmin, REAL tmax, int* brackt)
{
    /* arithmetic_ansi.h:34 (the non-IEEE branch is the one compiled) */
    int opposite = (dt * (best->d / REAL_ABS(best->d)) < 0.);
    int bound;
    REAL newt, mc, mq;

    if (*brackt) {
        REAL lo = best->st <= other->st ? best->st : other->st;
        REAL hi = best->st >= other->st ? best->st : other->st;
        if (*t <= lo || hi <= *t) return PLM_LBFGSERR_OUTOFINTERVAL;
        if (0. <= best->d * (*t - best->st)) return PLM_LBFGSERR_INCREASEGRADIENT;
        if (tmax < tmin) return PLM_LBFGSERR_INCORRECT_TMINMAX;
    }

    if (best->f < ft) {                               /* case 1: higher value */
        *brackt = 1; bound = 1;
        mc = FN(cubic_min)(best->st, best->f, best->d, *t, ft, dt);
        mq = FN(quad_min_f)(best->st, best->f, best->d, *t, ft);
        newt = (REAL_ABS(mc - best->st) < REAL_ABS(mq - best->st)) ? mc : mc + 0.5 * (mq - mc);
    } else if (opposite) {                            /* case 2: sign change */
        *brackt = 1; bound = 0;
        mc = FN(cubic_min)(best->st, best->f, best->d, *t, ft, dt);
        mq = FN(quad_min_d)(best->st, best->d, *t, dt);
        newt = (REAL_ABS(mc - *t) > REAL_ABS(mq - *t)) ? mc : mq;
    } else if (REAL_ABS(dt) < REAL_ABS(best->d)) {    /* case 3: derivative shrinks */
        bound = 1;
        mc = FN(cubic_min_clamped)(best->st, best->f, best->d, *t, ft, dt, tmin, tmax);
        mq = FN(quad_min_d)(best->st, best->d, *t, dt);
        if (*brackt) newt = (REAL_ABS(*t - mc) < REAL_ABS(*t - mq)) ? mc : mq;
        else         newt = (REAL_ABS(*t - mc) > REAL_ABS(*t - mq)) ? mc : mq;
    } else {                                          /* case 4 */
        bound = 0;
        if (*brackt) newt = FN(cubic_min)(*t, ft, dt, other->st, other->f, other->d);
        else if (best->st < *t) newt = tmax;
        else newt = tmin;
    }

    if (best->f < ft) {
        other->st = *t; other->f = ft; other->d = dt;
    } else {
        if (opposite) *other = *best;
        best->st = *t; best->f = ft; best->d = dt;
    }

    if (tmax < newt) newt = tmax;
    if (newt < tmin) newt = tmin;
    if (*brackt && bound) {
        mq = best->st + 0.66 * (other->st - best->st);
        if (best->st < other->st) { if (mq < newt) newt = mq; }
        else                      { if (newt < mq) newt = mq; }
    }
    *t = newt;
    return 0;
}

typedef REAL (*FN(eval_fn))(void* ctx, const REAL* x, REAL* g, size_t n);

/* line search, lbfgs.cpp:815-1004.  Returns #evaluations (>0) or an error code. */
static int FN(mt_search)(size_t n, REAL* x, REAL* f, REAL* g, const REAL* s, REAL* stp,
                         const REAL* xp, FN(eval_fn) eval, void* ctx, int* nevals)
{
    const REAL ftol = PLM_FTOL, gtol = PLM_GTOL, xtol = PLM_XTOL;
    const REAL min_step = PLM_MIN_STEP, max_step = PLM_MAX_STEP;
    const int max_ls = PLM_MAX_LINESEARCH;
    int count = 0, brackt = 0, stage1 = 1, uinfo = 0;
    REAL dg, finit, ftest1, dginit, dgtest, width, prev_width, stmin, stmax;
    FN(lspoint) bx, by;

    if (*stp <= 0.) return PLM_LBFGSERR_INVALIDPARAMETERS;
    dginit = FN(vdot)(g, s, n);
    if (0 < dginit) return PLM_LBFGSERR_INCREASEGRADIENT;

    finit = *f;
    dgtest = ftol * dginit;
    width = max_step - min_step;
    prev_width = 2.0 * width;
    bx.st = by.st = 0.; bx.f = by.f = finit; bx.d = by.d = dginit;

    for (;;) {
        if (brackt) {
            stmin = bx.st <= by.st ? bx.st : by.st;
            stmax = bx.st >= by.st ? bx.st : by.st;
        } else {
            stmin = bx.st;
            stmax = *stp + 4.0 * (*stp - bx.st);
        }
        if (*stp < min_step) *stp = min_step;
        if (max_step < *stp) *stp = max_step;
        if ((brackt && ((*stp <= stmin || stmax <= *stp) || max_ls <= count + 1 || uinfo != 0))
            || (brackt && (stmax - stmin <= xtol * stmax)))
            *stp = bx.st;

        for (size_t i = 0; i < n; ++i) { x[i] = xp[i]; x[i] += *stp * s[i]; }
        *f = eval(ctx, x, g, n);
        ++*nevals;
        dg = FN(vdot)(g, s, n);
        ftest1 = finit + *stp * dgtest;
        ++count;

        if (brackt && ((*stp <= stmin || stmax <= *stp) || uinfo != 0)) return PLM_LBFGSERR_ROUNDING_ERROR;
        if (*stp == max_step && *f <= ftest1 && dg <= dgtest) return PLM_LBFGSERR_MAXIMUMSTEP;
        if (*stp == min_step && (ftest1 < *f || dgtest <= dg)) return PLM_LBFGSERR_MINIMUMSTEP;
        if (brackt && (stmax - stmin) <= xtol * stmax) return PLM_LBFGSERR_WIDTHTOOSMALL;
        if (max_ls <= count) return PLM_LBFGSERR_MAXIMUMLINESEARCH;
        if (*f <= ftest1 && REAL_ABS(dg) <= gtol * (-dginit)) return count;

        if (stage1 && *f <= ftest1 && (ftol < gtol ? ftol : gtol) * dginit <= dg) stage1 = 0;

        if (stage1 && ftest1 < *f && *f <= bx.f) {
            /* work on the modified function psi(t) = f(t) - f(0) - ftol*t*f'(0) */
            FN(lspoint) mx, my;
            REAL fm = *f - *stp * dgtest, dgm = dg - dgtest;
            mx.st = bx.st; mx.f = bx.f - bx.st * dgtest; mx.d = bx.d - dgtest;
            my.st = by.st; my.f = by.f - by.st * dgtest; my.d = by.d - dgtest;
            uinfo = FN(mt_update)(&mx, &my, stp, fm, dgm, stmin, stmax, &brackt);
            bx.st = mx.st; bx.f = mx.f + mx.st * dgtest; bx.d = mx.d + dgtest;
            by.st = my.st; by.f = my.f + my.st * dgtest; by.d = my.d + dgtest;
        } else {
            uinfo = FN(mt_update)(&bx, &by, stp, *f, dg, stmin, stmax, &brackt);
        }

        if (brackt) {
            if (0.66 * prev_width <= REAL_ABS(by.st - bx.st)) *stp = bx.st + 0.5 * (by.st - bx.st);
            prev_width = width;
            width = REAL_ABS(by.st - bx.st);
        }
    }
}

typedef struct {
    const uint8_t* X; const REAL* w; int N, L, q; REAL lh, lJ; int carry, threads;
} FN(plm_ctx);

static REAL FN(plm_eval)(void* c, const REAL* x, REAL* g, size_t n)
{
    FN(plm_ctx)* p = (FN(plm_ctx)*)c; (void)n;
    return FN(oracle_gradient)(p->X, p->w, p->N, p->L, p->q, p->lh, p->lJ, x, g, p->carry, p->threads);
}

/* driver, lbfgs.cpp:248-644 with m=5, epsilon=1e-3 (plmdcaBackend.cpp:68-75).
 * x is in/out.  stats[0]=status, [1]=iterations completed, [2]=evaluations.
 * trace (optional, 4 REALs per iteration: fx,xnorm,gnorm,step) up to trace_cap iterations.
 * snap_iters / snap_x (optional): x after iteration snap_iters[s] is copied to snap_x + s*P (checkpoints of one long run). */
int FN(oracle_lbfgs)(const uint8_t* X, const REAL* w, int N, int L, int q,
                     REAL lambda_h, REAL lambda_J, int max_iterations, int carry, int threads,
                     REAL* x, REAL* fx_out, int* stats, REAL* trace, int trace_cap,
                     const int* snap_iters, int nsnap, REAL* snap_x)
{
    enum { M = PLM_LBFGS_M };
    const size_t n = plm_num_params(L, q);
    const REAL eps = PLM_EPSILON;
    FN(plm_ctx) ctx = { X, w, N, L, q, lambda_h, lambda_J, carry, threads };
    REAL *xp, *g, *gp, *d, *S[M], *Y[M], ysv[M], alpha[M];
    REAL fx, xnorm, gnorm, step, ys = 0, yy = 0, beta;
    int ret = 0, k = 1, end = 0, nevals = 0, iters = 0;

    xp = (REAL*)calloc(n, sizeof(REAL)); g = (REAL*)calloc(n, sizeof(REAL));
    gp = (REAL*)calloc(n, sizeof(REAL)); d = (REAL*)calloc(n, sizeof(REAL));
    for (int i = 0; i < M; ++i) {
        S[i] = (REAL*)calloc(n, sizeof(REAL)); Y[i] = (REAL*)calloc(n, sizeof(REAL));
        ysv[i] = 0; alpha[i] = 0;
    }

    fx = FN(plm_eval)(&ctx, x, g, n); ++nevals;
    for (size_t i = 0; i < n; ++i) d[i] = -g[i];
    xnorm = REAL_SQRT(FN(vdot)(x, x, n));
    gnorm = REAL_SQRT(FN(vdot)(g, g, n));
    if (xnorm < 1.0) xnorm = 1.0;
    if (gnorm / xnorm <= eps) { ret = PLM_LBFGS_ALREADY_MINIMIZED; goto done; }
    step = (REAL)(1.0 / REAL_SQRT(FN(vdot)(d, d, n)));

    for (;;) {
        memcpy(xp, x, n * sizeof(REAL));
        memcpy(gp, g, n * sizeof(REAL));
        int ls = FN(mt_search)(n, x, &fx, g, d, &step, xp, FN(plm_eval), &ctx, &nevals);
        if (ls < 0) {
            memcpy(x, xp, n * sizeof(REAL));
            memcpy(g, gp, n * sizeof(REAL));
            ret = ls;
            break;
        }
        xnorm = REAL_SQRT(FN(vdot)(x, x, n));
        gnorm = REAL_SQRT(FN(vdot)(g, g, n));
        iters = k;
        if (trace && k <= trace_cap) {
            REAL* t = trace + (size_t)(k - 1) * 4;
            t[0] = fx; t[1] = xnorm; t[2] = gnorm; t[3] = step;
        }
        for (int sidx = 0; sidx < nsnap; ++sidx)
            if (snap_iters[sidx] == k) memcpy(snap_x + (size_t)sidx * n, x, n * sizeof(REAL));
        if (getenv("ORACLE_PROGRESS_FILE")) {      /* long runs (config D: half a minute per evaluation): one line per iteration */
            FILE* pf = fopen(getenv("ORACLE_PROGRESS_FILE"), "a");
            if (pf) { fprintf(pf, "%d %.17g %.17g %.17g %.17g %d\n", k, (double)fx, (double)xnorm, (double)gnorm, (double)step, nevals); fclose(pf); }
        }
        if (xnorm < 1.0) xnorm = 1.0;
        if (gnorm / xnorm <= eps) { ret = PLM_LBFGS_SUCCESS; break; }
        if (max_iterations != 0 && max_iterations < k + 1) { ret = PLM_LBFGSERR_MAXIMUMITERATION; break; }

        for (size_t i = 0; i < n; ++i) { S[end][i] = x[i] - xp[i]; Y[end][i] = g[i] - gp[i]; }
        ys = FN(vdot)(Y[end], S[end], n);
        yy = FN(vdot)(Y[end], Y[end], n);
        ysv[end] = ys;

        int bound = (M <= k) ? M : k;
        ++k;
        end = (end + 1) % M;
        for (size_t i = 0; i < n; ++i) d[i] = -g[i];
        int j = end;
        for (int i = 0; i < bound; ++i) {
            j = (j + M - 1) % M;
            alpha[j] = FN(vdot)(S[j], d, n);
            alpha[j] /= ysv[j];
            REAL c = -alpha[j];
            for (size_t t = 0; t < n; ++t) d[t] += c * Y[j][t];
        }
        { REAL c = ys / yy; for (size_t t = 0; t < n; ++t) d[t] *= c; }
        for (int i = 0; i < bound; ++i) {
            beta = FN(vdot)(Y[j], d, n);
            beta /= ysv[j];
            REAL c = alpha[j] - beta;
            for (size_t t = 0; t < n; ++t) d[t] += c * S[j][t];
            j = (j + 1) % M;
        }
        step = 1.0;
    }

done:
    if (fx_out) *fx_out = fx;
    if (stats) { stats[0] = ret; stats[1] = iters; stats[2] = nevals; }
    free(xp); free(g); free(gp); free(d);
    for (int i = 0; i < M; ++i) { free(S[i]); free(Y[i]); }
    return ret;
}
