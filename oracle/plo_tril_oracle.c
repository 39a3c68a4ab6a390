/* ==========================================================================
 * oracle/plo_tril_oracle.c -- TEST INFRASTRUCTURE ONLY (see plo_oracle.h).
 *
 * Plain-C restatement of PLinOpt's in-place trilinear search (trilplacer):
 *   Atom / cumulate / isnoop          include/plinopt_inplace.inl:15-124
 *   complexity                        :133-144
 *   orientindex / nextindex           :179-236
 *   simplify                          :243-311
 *   pushvariables                     :322-393
 *   LinearAlgorithm                   :400-502
 *   TransposedDoubleAlgorithm         :507-598   (`trilplacer -e`; DoubleExpand :676-716 done on the fly)
 *   TriLinearProgram                  :732-806   (text: Atom operator<< :43-78, macros plinopt_inplace.h:90-112)
 *   SearchTriLinearAlgorithm          :812-929   (one restart = one candidate seed here)
 * over the rationals (int64 numerator/denominator, overflow-checked).
 *
 * PARITY STATUS: "parity unpinned" against the genuine binary (needs LinBox/Givaro; its own check is a Maple script
 * behind -DINPLACE_CHECKER).  Random choices are the build's frozen per-candidate stream (plo_oracle.h): Fisher-Yates
 * row permutation (LinBox Permutation::random is not in the tree), one bit per brand(), `next % n` for the variable
 * picks.  What IS pinned (tests/test_tril_oracle.py): every emitted program, run by an independent in-place
 * interpreter on random rational inputs, adds the bilinear map to c and restores a and b; printed counts equal the
 * operation count of the text.
 * ========================================================================== */
#include "plo_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>

/* ---------------------------------------------------------------- rationals */
/* 128-bit numerator and denominator, overflow-checked (the reference's Givaro::Rational is arbitrary precision: the fixtures with
 * decimal coefficients such as 12.0695 overflow 64 bits in cumulate :96-122 after a few merges, not 128) */
typedef __int128 i128;
typedef struct { i128 n, d; } rat;
static int g_overflow;
static i128 gcd128(i128 a, i128 b) { if (a < 0) a = -a; if (b < 0) b = -b; while (b) { i128 t = a % b; a = b; b = t; } return a; }
static rat rmake(i128 n, i128 d) {
    if (d == 0) { g_overflow = 1; rat z = {0, 1}; return z; }
    if (d < 0) { n = -n; d = -d; }
    i128 g = gcd128(n, d); if (g > 1) { n /= g; d /= g; }
    rat r = {n, d}; return r;
}
static i128 mulck(i128 a, i128 b) { i128 r; if (__builtin_mul_overflow(a, b, &r)) g_overflow = 1; return r; }
static i128 addck(i128 a, i128 b) { i128 r; if (__builtin_add_overflow(a, b, &r)) g_overflow = 1; return r; }
static rat radd(rat a, rat b) { const i128 g = gcd128(a.d, b.d); const i128 bd = b.d / g, ad = a.d / g; return rmake(addck(mulck(a.n, bd), mulck(b.n, ad)), mulck(a.d, bd)); }
static rat rneg(rat a) { rat r = {-a.n, a.d}; return r; }
static rat rsub(rat a, rat b) { return radd(a, rneg(b)); }
static rat rmul(rat a, rat b) { const i128 g1 = gcd128(a.n, b.d), g2 = gcd128(b.n, a.d); return rmake(mulck(a.n / (g1 ? g1 : 1), b.n / (g2 ? g2 : 1)), mulck(a.d / (g2 ? g2 : 1), b.d / (g1 ? g1 : 1))); }
static rat rinv(rat a) { return rmake(a.d, a.n); }
static rat rdiv(rat a, rat b) { return rmul(a, rinv(b)); }
static int rsign(rat a) { return a.n > 0 ? 1 : a.n < 0 ? -1 : 0; }
static int risone(rat a) { return a.n == 1 && a.d == 1; }
static int rismone(rat a) { return a.n == -1 && a.d == 1; }
static int riszero(rat a) { return a.n == 0; }
static int rabslt1(rat a) { i128 n = a.n < 0 ? -a.n : a.n; return n < a.d; }      /* |a| < 1 */

/* -------------------------------------------------------------------- atoms */
typedef struct { char var; uint32_t src; char ope; rat val; long des; } atom;   /* plinopt_inplace.inl:15-24 */
typedef struct { atom *a; size_t n, cap; } prog;

static void ppush(prog *P, char var, uint32_t src, char ope, rat val, long des) {
    if (P->n == P->cap) { P->cap = P->cap ? 2 * P->cap : 64; P->a = (atom *)realloc(P->a, P->cap * sizeof(atom)); }
    atom t = {var, src, ope, val, des}; P->a[P->n++] = t;
}
static void perase(prog *P, size_t k) { memmove(P->a + k, P->a + k + 1, (P->n - k - 1) * sizeof(atom)); --P->n; }
static int is_addsub(char c) { return c == '+' || c == '-'; }            /* plinopt_inplace.h:82-83 */
static int is_muldiv(char c) { return c == '*' || c == '/'; }
static char swapop(char c) { return c == '+' ? '-' : '+'; }              /* :90-92 */
static char invop(char c) { return c == '*' ? '/' : '*'; }
static char moneop(char op, rat v) { return rismone(v) ? swapop(op) : op; }

static int sameops(const atom *a, const atom *b) { return a->var == b->var && a->src == b->src && a->des == b->des; }   /* :81-85 */
static int isnoop(const atom *a) { return (is_addsub(a->ope) && riszero(a->val)) || (is_muldiv(a->ope) && risone(a->val)); }   /* :88-91 */
static int cumulate(atom *t, const atom *p) {                           /* :94-122 */
    if (!sameops(t, p)) return 0;
    if (is_addsub(t->ope) && is_addsub(p->ope)) {
        t->val = (t->ope == p->ope) ? radd(t->val, p->val) : rsub(t->val, p->val);
        if (rsign(t->val) < 0) { t->ope = swapop(t->ope); t->val = rneg(t->val); }
        return 1;
    }
    if (is_muldiv(t->ope) && is_muldiv(p->ope)) {
        t->val = (t->ope == p->ope) ? rmul(t->val, p->val) : rdiv(t->val, p->val);
        if (rabslt1(t->val)) { t->ope = invop(t->ope); t->val = rinv(t->val); }
        return 1;
    }
    return 0;
}

/* :133-144 */
static void complexity(const prog *P, uint32_t out[3]) {
    out[0] = out[1] = out[2] = 0;
    for (size_t k = 0; k < P->n; ++k) {
        const atom *t = &P->a[k];
        if (is_addsub(t->ope)) { ++out[0]; if (!risone(t->val) && !rismone(t->val)) ++out[1]; }
        if (is_muldiv(t->ope)) ++out[1];
        if (t->ope == ' ') ++out[2];
    }
}

/* :243-311: merge the first atom that meets a later compatible one before a dependency intervenes */
static int simplify(prog *P, int transposed) {
    for (size_t i = 0; i < P->n; ++i) {
        const atom *it = &P->a[i];
        if (it->ope == ' ') continue;
        for (size_t k = i + 1; k < P->n; ++k) {
            const atom *nx = &P->a[k];
            if (sameops(nx, it)) {
                atom c = *it;
                if (cumulate(&c, nx)) {
                    perase(P, k);
                    if (isnoop(&c)) perase(P, i); else P->a[i] = c;
                    return 1;
                }
            }
            int needbreak = 0;
            needbreak |= (it->src == nx->src) && (nx->ope == ' ' || (is_addsub(it->ope) && is_muldiv(nx->ope)) || (is_muldiv(it->ope) && is_addsub(nx->ope)));
            if (transposed) needbreak |= (it->des == (long)nx->src);
            else needbreak |= (it->des == (long)nx->src) && (nx->ope != ' ');
            needbreak |= ((long)it->src == nx->des);
            if (needbreak) break;
        }
    }
    return 0;
}

static void rotate_left1(prog *P, size_t f, size_t e) {   /* std::rotate(findex, findex+1, iter): a[f] goes to e-1 */
    atom t = P->a[f];
    memmove(P->a + f, P->a + f + 1, (e - f - 1) * sizeof(atom));
    P->a[e - 1] = t;
}
/* :322-393 */
static void pushvariables(prog *P, size_t numout) {
    for (size_t i = 0; i < numout; ++i) {
        int found = 0; size_t f = 0;
        for (size_t k = 0; k < P->n; ++k) {
            const atom *it = &P->a[k];
            if (found) {
                const atom *fx = &P->a[f];
                int rot = 0;
                if (is_addsub(fx->ope)) {
                    if (fx->des == (long)it->src) { found = 0; continue; }
                    else if (it->src == i) {
                        if (fx->des == it->des) rot = 1;
                        else if (is_muldiv(it->ope)) { found = 0; continue; }
                    }
                } else {
                    if (it->des == (long)i) { found = 0; continue; }
                    else if (it->src == i) {
                        if (is_muldiv(it->ope)) rot = 1; else { found = 0; continue; }
                    }
                }
                if (rot) {
                    if (f + 1 != k) rotate_left1(P, f, k);
                    found = 0;
                }
            } else if (it->ope != ' ' && it->src == i) { found = 1; f = k; }
        }
        if (found && f + 1 != P->n) rotate_left1(P, f, P->n);
    }
}

/* matrices: CSR, rational values (den == NULL: integers) */
typedef struct { uint32_t m, n; const uint32_t *rp, *col; const int64_t *num, *den; } mat;
typedef struct { uint32_t col; rat v; } ent;

/* :179-236 with the build's stream */
static size_t nextindex(size_t preci, const ent *L, size_t len, int oriented, uint32_t *rng) {
    if (!oriented) return plo_oracle_rng_next(rng) % (uint32_t)len;              /* :226-232 */
    size_t nexti = len;
    for (size_t k = 0; k < len; ++k) if (L[k].col == preci) { nexti = k; break; }
    if (nexti == len || !risone(L[nexti].v)) {
        size_t cnt = 0;
        for (size_t k = 0; k < len; ++k) if (risone(L[k].v)) ++cnt;
        if (cnt > 0) {
            uint32_t pick = plo_oracle_rng_next(rng) % (uint32_t)cnt;             /* :199-201 */
            for (size_t k = 0; k < len; ++k) if (risone(L[k].v)) { if (pick == 0) { nexti = k; break; } --pick; }
        }
    }
    return nexti != len ? nexti : 0;
}

/* :400-502; rows given through perm/sign: row l of the permuted matrix = sign[l] * M[perm[l]] */
static void linear_algorithm(prog *P, const mat *M, const uint32_t *perm, const int8_t *sgn, char variable, int transposed,
                             int oriented, uint32_t *rng, uint32_t ops[3]) {
    size_t preci = M->n;
    ent *L = (ent *)malloc((M->n + 1) * sizeof(ent));
    for (uint32_t l = 0; l < M->m; ++l) {
        const uint32_t r = perm ? perm[l] : l;
        size_t len = 0;
        for (uint32_t e = M->rp[r]; e < M->rp[r + 1]; ++e) {
            rat v = rmake(M->num[e], M->den ? M->den[e] : 1);
            if (sgn && sgn[l] < 0) v = rneg(v);
            L[len].col = M->col[e]; L[len].v = v; ++len;
        }
        if (len > 0) {
            const size_t ai = nextindex(preci, L, len, oriented, rng);
            const uint32_t i = L[ai].col; const rat av = L[ai].v;
            if (!risone(av)) {
                if (transposed) { if (!rismone(av)) ppush(P, variable, i, '/', av, -1); }
                else ppush(P, variable, i, '*', av, -1);
            }
            for (size_t k = 0; k < len; ++k) if (k != ai) {
                if (transposed) ppush(P, variable, L[k].col, moneop('-', av), L[k].v, (long)i);
                else ppush(P, variable, i, '+', L[k].v, (long)L[k].col);
            }
            ppush(P, variable, i, ' ', av, -1);
            for (size_t k = 0; k < len; ++k) if (k != ai) {
                if (transposed) ppush(P, variable, L[k].col, moneop('+', av), L[k].v, (long)i);
                else ppush(P, variable, i, '-', L[k].v, (long)L[k].col);
            }
            if (!risone(av)) {
                if (transposed) { if (!rismone(av)) ppush(P, variable, i, '*', av, -1); }
                else ppush(P, variable, i, '/', av, -1);
            }
            if (len > 1) preci = i;
        } else {
            rat z = {0, 1}; ppush(P, ' ', l, ' ', z, -1);
        }
    }
    free(L);
    {   /* remove_if(isMulDivOne) :481-482 */
        size_t w = 0;
        for (size_t k = 0; k < P->n; ++k) if (!(is_muldiv(P->a[k].ope) && risone(P->a[k].val))) P->a[w++] = P->a[k];
        P->n = w;
    }
    int simp;
    do { if (transposed) pushvariables(P, M->n); simp = simplify(P, transposed); } while (simp);
    complexity(P, ops);
}

/* TransposedDoubleAlgorithm :507-598 on TT = DoubleExpand(T) (:676-716: row 2l of TT is row l of T, row 2l+1 the same
 * entries one column to the right; TT has n+1 columns).  Rows through perm/sign as in SearchTriLinearAlgorithm :846-883:
 * row ll of the permuted TT = sign[ll>>1] * TT[2*perm[ll>>1] + (ll&1)]. */
static int notabsone(rat a) { return !risone(a) && !rismone(a); }
static void transposed_double_algorithm(prog *P, const mat *T, const uint32_t *perm, const int8_t *sgn, char variable, uint32_t ops[3]) {
    ent *L = (ent *)malloc((T->n + 1) * sizeof(ent));
    const rat zero = {0, 1};
    /* One trip per PAIR of rows (2l, 2l+1) of TT: the trip works on the block <<a|c>,<0|a>> of :519-531, whose lower row is
     * the upper one moved one column to the right, and emits the two barriers that one MULTD consumes.  The reference's
     * loop header (:515) reads `++l`: taken literally every odd row would be worked on a second time as an upper row, the
     * program would hold 4m barriers for m AXPYs and the synchronisation loop of TriLinearProgram (:756-783) could not end
     * (it cannot pass a barrier of c once the barriers of a are used up).  Stepping by two is what the block structure,
     * MULTD and the Maple check (:1047-1061) describe; the tests hold the programs to that check. */
    for (uint32_t ll = 0; ll < 2u * T->m; ll += 2) {
        const uint32_t l = ll >> 1, sh = 0, r = perm ? perm[l] : l;
        size_t len = 0;
        for (uint32_t e = T->rp[r]; e < T->rp[r + 1]; ++e) {
            rat v = rmake(T->num[e], T->den ? T->den[e] : 1);
            if (sgn && sgn[l] < 0) v = rneg(v);
            L[len].col = T->col[e] + sh; L[len].v = v; ++len;
        }
        if (len > 0) {
            const uint32_t i = L[0].col, cindex = i + 1;                      /* matrix <<a|c>,<0|a>> */
            const rat a = L[0].v, y = rinv(a);
            rat c = zero, z = zero;
            if (len > 1 && L[1].col == cindex) { c = L[1].v; z = rneg(rmul(rmul(y, c), y)); }   /* z = - a^-1 c a^-1 */
            if (notabsone(y)) ppush(P, variable, cindex, '*', y, -1);
            if (!riszero(z)) ppush(P, variable, cindex, moneop('+', y), z, (long)i);
            if (notabsone(y)) ppush(P, variable, i, '*', y, -1);
            for (size_t k = 1; k < len; ++k) {
                if (L[k].col != cindex) ppush(P, variable, L[k].col, moneop('-', y), L[k].v, (long)i);
                ppush(P, variable, L[k].col + 1, moneop('-', y), L[k].v, (long)cindex);
            }
            ppush(P, variable, i, ' ', a, -1);
            ppush(P, variable, cindex, ' ', a, -1);
            for (size_t k = 1; k < len; ++k) {
                if (L[k].col != cindex) ppush(P, variable, L[k].col, moneop('+', a), L[k].v, (long)i);
                ppush(P, variable, L[k].col + 1, moneop('+', a), L[k].v, (long)cindex);
            }
            if (notabsone(a)) ppush(P, variable, cindex, '*', a, -1);
            if (!riszero(c)) ppush(P, variable, cindex, moneop('+', a), c, (long)i);
            if (notabsone(a)) ppush(P, variable, i, '*', a, -1);
        } else {
            ppush(P, ' ', ll, ' ', zero, -1);
        }
    }
    free(L);
    {   /* remove_if(isMulDivOne) :586-587 */
        size_t w = 0;
        for (size_t k = 0; k < P->n; ++k) if (!(is_muldiv(P->a[k].ope) && risone(P->a[k].val))) P->a[w++] = P->a[k];
        P->n = w;
    }
    int simp;
    do { pushvariables(P, (size_t)T->n + 1); simp = simplify(P, 1); } while (simp);
    complexity(P, ops);
}

/* ------------------------------------------------------------------- text */
typedef struct { char *s; size_t n, cap; } sbuf;
static void sput(sbuf *b, const char *fmt, ...) {
    va_list ap; char tmp[256];
    va_start(ap, fmt); int k = vsnprintf(tmp, sizeof tmp, fmt, ap); va_end(ap);
    if (b->n + (size_t)k + 1 > b->cap) { b->cap = 2 * (b->cap + (size_t)k) + 256; b->s = (char *)realloc(b->s, b->cap); }
    memcpy(b->s + b->n, tmp, (size_t)k + 1); b->n += (size_t)k;
}
static void put_i128(sbuf *b, i128 v) {
    char t[48]; int k = 47; t[k] = 0; const int neg = v < 0; unsigned __int128 u = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
    do { t[--k] = (char)('0' + (int)(u % 10)); u /= 10; } while (u);
    if (neg) t[--k] = '-';
    sput(b, "%s", t + k);
}
static void put_rat(sbuf *b, rat r) { put_i128(b, r.n); if (r.d != 1) { sput(b, "/"); put_i128(b, r.d); } }
/* printmulorjustdiv, rational specialisation: plinopt_library.inl:360-374 */
static void put_mulordiv(sbuf *b, char c, long i, rat r) {
    sput(b, "%c%ld", c, i);
    if (!risone(r)) { if (r.n == 1) { sput(b, "/"); put_i128(b, r.d); } else { sput(b, "*"); put_rat(b, r); } }
}
/* Atom operator<< :43-78 (barriers are consumed by the AXPY lines of the caller) */
static void put_atom(sbuf *b, const atom *p) {
    const int bsca = is_muldiv(p->ope);
    if (bsca && risone(p->val)) return;
    if (p->ope == ' ') { if (riszero(p->val)) sput(b, "0;"); else sput(b, "%c%u;", p->var, p->src); sput(b, "\n"); return; }
    sput(b, "%c%u:=", p->var, p->src);
    const rat uval = rsign(p->val) < 0 ? rneg(p->val) : p->val;
    if (bsca) {
        if (rsign(p->val) < 0) sput(b, "-");
        put_mulordiv(b, p->var, (long)p->src, p->ope == '*' ? uval : rinv(uval));          /* printSCA :376-386 */
    } else {
        const char uope = rsign(p->val) < 0 ? swapop(p->ope) : p->ope;
        sput(b, "%c%u%c", p->var, p->src, uope);
        put_mulordiv(b, p->var, p->des, uval);
    }
    sput(b, ";\n");
}

/* :732-806; expanded (`-e`): the c program is TransposedDoubleAlgorithm on DoubleExpand(T), an AXPY takes two barriers of c (MULTD) */
static void trilinear_program(sbuf *out, const mat *A, const mat *B, const mat *T, const uint32_t *perm, const int8_t *sa,
                              const int8_t *sb, const int8_t *st, int oriented, int expanded, uint32_t *rng, uint32_t nops[3]) {
    prog pa = {0, 0, 0}, pb = {0, 0, 0}, pc = {0, 0, 0};
    uint32_t oa[3], ob[3], oc[3];
    linear_algorithm(&pa, A, perm, sa, 'a', 0, oriented, rng, oa);
    linear_algorithm(&pb, B, perm, sb, 'b', 0, oriented, rng, ob);
    if (expanded) transposed_double_algorithm(&pc, T, perm, st, 'c', oc);
    else linear_algorithm(&pc, T, perm, st, 'c', 1, oriented, rng, oc);
    if (out) {
        sput(out, "# Found %u|%u|%u for a\n# Found %u|%u|%u for b\n# Found %u|%u|%u for c\n", oa[0], oa[1], oa[2], ob[0], ob[1], ob[2], oc[0], oc[1], oc[2]);
        size_t ia = 0, ib = 0, ic = 0;
        while (ic < pc.n) {
            for (; ia < pa.n && pa.a[ia].ope != ' '; ++ia) put_atom(out, &pa.a[ia]);
            for (; ib < pb.n && pb.a[ib].ope != ' '; ++ib) put_atom(out, &pb.a[ib]);
            for (; ic < pc.n && pc.a[ic].ope != ' '; ++ic) put_atom(out, &pc.a[ic]);
            if (ia < pa.n && ib < pb.n && ic < pc.n) {       /* MUL / MULTD macros, plinopt_inplace.h:108-111 */
                const atom *c = &pc.a[ic];
                if (expanded) {
                    const atom *c2 = ic + 1 < pc.n ? &pc.a[ic + 1] : c;
                    sput(out, "%c%u:=%c%u %c (%c%u * %c%u)*low; ### AXPY low  ###\n", c->var, c->src, c->var, c->src, moneop('+', c->val),
                         pa.a[ia].var, pa.a[ia].src, pb.a[ib].var, pb.a[ib].src);
                    sput(out, "%c%u:=%c%u %c (%c%u * %c%u)*hig; ### AXPY high ###\n", c2->var, c2->src, c2->var, c2->src, moneop('+', c->val),
                         pa.a[ia].var, pa.a[ia].src, pb.a[ib].var, pb.a[ib].src);
                    ++ic;
                } else
                sput(out, "%c%u:=%c%u %c %c%u * %c%u; ### AXPY ###\n", c->var, c->src, c->var, c->src, moneop('+', c->val),
                     pa.a[ia].var, pa.a[ia].src, pb.a[ib].var, pb.a[ib].src);
                ++ia; ++ib; ++ic;
            }
        }
        for (; ic < pc.n; ++ic) put_atom(out, &pc.a[ic]);
        for (; ib < pb.n; ++ib) put_atom(out, &pb.a[ib]);
        for (; ia < pa.n; ++ia) put_atom(out, &pa.a[ia]);
    }
    if (expanded) oc[2] >>= 1;                                   /* :799: MUL2D is counted twice */
    nops[0] = oa[0] + ob[0] + oc[0]; nops[1] = oa[1] + ob[1] + oc[1]; nops[2] = (oa[2] + ob[2] + oc[2]) / 3u;
    free(pa.a); free(pb.a); free(pc.a);
}

/* one restart of :837-924: permutation, coherent negations, oriented then unoriented program.
 * seed == PLO_TRIL_BASE (all ones): the unpermuted oriented program of :829 only (variant 1 = copy of variant 0). */
static int tril_candidate(const mat *A, const mat *B, const mat *T, int expanded, uint64_t seed, uint32_t ops[6], int want_variant, char **text) {
    const uint32_t m = A->m;
    uint32_t rng = plo_oracle_rng_state0(seed);
    uint32_t *perm = (uint32_t *)malloc(m * sizeof(uint32_t));
    int8_t *sa = (int8_t *)malloc(m), *sb = (int8_t *)malloc(m), *st = (int8_t *)malloc(m);
    for (uint32_t i = 0; i < m; ++i) { perm[i] = i; sa[i] = sb[i] = st[i] = 1; }
    const int base = (seed == ~0ull);
    if (!base) {
        for (uint32_t i = m; i > 1; --i) { uint32_t j = plo_oracle_rng_next(&rng) % i; uint32_t t = perm[i - 1]; perm[i - 1] = perm[j]; perm[j] = t; }
        for (uint32_t i = 0; i < m; ++i) {                       /* :872-885 */
            const int na = (int)(plo_oracle_rng_next(&rng) & 1u), nb = (int)(plo_oracle_rng_next(&rng) & 1u);
            if (na) sa[i] = -1;
            if (nb) sb[i] = -1;
            if (na != nb) st[i] = -1;
        }
    }
    g_overflow = 0;
    sbuf b0 = {0, 0, 0}, b1 = {0, 0, 0};
    trilinear_program(text && (want_variant == 0 || base) ? &b0 : NULL, A, B, T, perm, sa, sb, st, 1, expanded, &rng, ops);
    if (base) { ops[3] = ops[0]; ops[4] = ops[1]; ops[5] = ops[2]; }
    else trilinear_program(text && want_variant == 1 ? &b1 : NULL, A, B, T, perm, sa, sb, st, 0, expanded, &rng, ops + 3);
    if (text) { *text = want_variant == 0 || base ? b0.s : b1.s; if (!*text) { *text = (char *)malloc(1); (*text)[0] = 0; } }
    free(perm); free(sa); free(sb); free(st);
    return g_overflow ? -3 : 0;
}

static int check_mats(const mat *A, const mat *B, const mat *T) { return (A->m == B->m && A->m == T->m && A->m > 0) ? 0 : -1; }

#define MAT(x_) { m, n##x_, rp##x_, col##x_, num##x_, den##x_ }
#define TRIL_PARAMS uint32_t m, \
    uint32_t nA, const uint32_t *rpA, const uint32_t *colA, const int64_t *numA, const int64_t *denA, \
    uint32_t nB, const uint32_t *rpB, const uint32_t *colB, const int64_t *numB, const int64_t *denB, \
    uint32_t nT, const uint32_t *rpT, const uint32_t *colT, const int64_t *numT, const int64_t *denT
#define TRIL_PASS m, nA, rpA, colA, numA, denA, nB, rpB, colB, numB, denB, nT, rpT, colT, numT, denT

/* expanded != 0: `trilplacer -e` (T is the m-row matrix; its double expansion is built inside) */
int plo_oracle_tril_cost_many_x(TRIL_PARAMS, int expanded, const uint64_t *seeds, uint64_t seed0, uint64_t nseeds, uint32_t *ops6) {
    const mat A = MAT(A), B = MAT(B), T = MAT(T);
    if (check_mats(&A, &B, &T)) return -1;
    int rc = 0;                             /* g_overflow is a plain global: the checker is serial */
    for (uint64_t k = 0; k < nseeds; ++k) { int r = tril_candidate(&A, &B, &T, expanded, seeds ? seeds[k] : seed0 + k, ops6 + 6 * k, 0, NULL); if (r) rc = r; }
    return rc;
}
int plo_oracle_tril_program_x(TRIL_PARAMS, int expanded, uint64_t seed, int variant, uint32_t *ops6, char **text) {
    const mat A = MAT(A), B = MAT(B), T = MAT(T);
    if (check_mats(&A, &B, &T)) return -1;
    return tril_candidate(&A, &B, &T, expanded, seed, ops6, variant, text);
}
/* best over [seed0, seed0+nseeds) x {oriented, unoriented} under (ADD, SCA, seed, variant); :893-923 */
int plo_oracle_tril_search_x(TRIL_PARAMS, int expanded, uint64_t seed0, uint64_t nseeds, uint32_t *best_ops3, uint64_t *best_seed, uint32_t *best_variant) {
    const mat A = MAT(A), B = MAT(B), T = MAT(T);
    if (check_mats(&A, &B, &T)) return -1;
    int have = 0;
    for (uint64_t k = 0; k < nseeds; ++k) {
        uint32_t o[6];
        int r = tril_candidate(&A, &B, &T, expanded, seed0 + k, o, 0, NULL);
        if (r) return r;
        for (uint32_t v = 0; v < 2; ++v) {
            const uint32_t *q = o + 3 * v;
            if (!have || q[0] < best_ops3[0] || (q[0] == best_ops3[0] && q[1] < best_ops3[1])) {
                have = 1; best_ops3[0] = q[0]; best_ops3[1] = q[1]; best_ops3[2] = q[2]; *best_seed = seed0 + k; *best_variant = v;
            }
        }
    }
    return have ? 0 : -1;
}
int plo_oracle_tril_cost_many(TRIL_PARAMS, const uint64_t *seeds, uint64_t seed0, uint64_t nseeds, uint32_t *ops6, int nthreads) {
    (void)nthreads;
    return plo_oracle_tril_cost_many_x(TRIL_PASS, 0, seeds, seed0, nseeds, ops6);
}
int plo_oracle_tril_program(TRIL_PARAMS, uint64_t seed, int variant, uint32_t *ops6, char **text) { return plo_oracle_tril_program_x(TRIL_PASS, 0, seed, variant, ops6, text); }
int plo_oracle_tril_search(TRIL_PARAMS, uint64_t seed0, uint64_t nseeds, uint32_t *best_ops3, uint64_t *best_seed, uint32_t *best_variant) {
    return plo_oracle_tril_search_x(TRIL_PASS, 0, seed0, nseeds, best_ops3, best_seed, best_variant);
}
