/*
 * oracle/sru_oracle.c -- TEST INFRASTRUCTURE ONLY (not product code).
 *
 * CPU restatement, in plain C, of the SRU (Simple Recurrent Unit, v1) element-wise
 * recurrence that the reference reaches through its un-vendored third-party
 * dependency `sru` (reference call sites: dmn_pytorch/models/dmn.py:7,104,120,156,204;
 * pinned at taolei87/sru@43c85ed by README.md:33).  That package is NOT present in
 * /root/reference nor anywhere in this container, so this file restates the published
 * algorithm (Lei et al., "Simple Recurrent Units for Highly Parallelizable Recurrence",
 * arXiv:1709.02755, eqs. 3-7, and the per-column kernel form recorded in SURVEY.md
 * section 2.1):
 *
 *     U  = x . W          viewed (T, B, d, k), k innermost          [done by the caller]
 *     f  = sigmoid(U[..,1] + b_f)          b_f = bias[0:d]
 *     r  = sigmoid(U[..,2] + b_r)          b_r = bias[d:2d]
 *     c_t = (c_{t-1} - U[..,0]) * f + U[..,0]
 *     x'  = x[t,b,j] * scale_x   (k == 3)   |   U[..,3]   (k == 4)
 *     h_t = (act(c_t) - x') * r + x'
 *
 * PARITY UNPINNED: the reference tree holds no golden vector / known-answer test for
 * this arithmetic (its only test checks an output *shape*, models/tests/test_dmn.py:73),
 * and the upstream source cannot be fetched.  Uncertain upstream details are explicit
 * arguments here (act, scale_x) exactly as in the HIP kernels.  The hand-derived
 * backward below is cross-checked in tests/ against torch autograd of the same
 * equations (oracle/sru_ref.py), i.e. two independent restatements must agree.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 *
 * Loop order is time-outer / column-inner (cache friendly on a CPU); the upstream GPU
 * kernel is column-outer / time-inner.  Every column's arithmetic is independent so
 * the results are identical.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

static inline float act_fwd(int act, float v)
{
    if (act == 1) return tanhf(v);
    if (act == 2) return v > 0.0f ? v : 0.0f;
    return v;
}

/* derivative of the activation expressed through its OUTPUT value y = act(v) */
static inline float act_bwd(int act, float y)
{
    if (act == 1) return 1.0f - y * y;
    if (act == 2) return y > 0.0f ? 1.0f : 0.0f;
    return 1.0f;
}

int sru_oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/*
 * u      (T, B, d, k)   x (T, B, d) or NULL when k == 4
 * bias   (2*d)          c0 (B, d) or NULL (zeros)
 * h      (T, B, d)      c  (T, B, d) or NULL       c_last (B, d) or NULL
 */
int sru_oracle_fwd(const float *u, const float *x, const float *bias, const float *c0,
                   int T, int B, int d, int k, int act, float scale_x,
                   float *h, float *c, float *c_last)
{
    if (!(k == 3 || k == 4) || T < 0 || B <= 0 || d <= 0) return -1;
    if (k == 3 && x == NULL) return -2;
    const ptrdiff_t ncols = (ptrdiff_t)B * d;
    float *cur = (float *)malloc(sizeof(float) * (size_t)ncols);
    if (!cur) return -3;
    if (c0) memcpy(cur, c0, sizeof(float) * (size_t)ncols);
    else    memset(cur, 0, sizeof(float) * (size_t)ncols);

#pragma omp parallel
    {
        for (int t = 0; t < T; ++t) {
            const float *ut = u + (ptrdiff_t)t * ncols * k;
            const float *xt = x ? x + (ptrdiff_t)t * ncols : NULL;
            float *ht = h + (ptrdiff_t)t * ncols;
            float *ct = c ? c + (ptrdiff_t)t * ncols : NULL;
#pragma omp for schedule(static)
            for (ptrdiff_t col = 0; col < ncols; ++col) {
                const int j = (int)(col % d);
                const float *up = ut + col * k;
                const float u0 = up[0];
                const float g1 = sigm(up[1] + bias[j]);
                const float g2 = sigm(up[2] + bias[j + d]);
                const float cv = (cur[col] - u0) * g1 + u0;
                cur[col] = cv;
                if (ct) ct[col] = cv;
                const float xv = (k == 3) ? xt[col] * scale_x : up[3];
                ht[col] = (act_fwd(act, cv) - xv) * g2 + xv;
            }
            /* implicit barrier of `omp for` orders step t before t+1 for every column */
        }
    }
    if (c_last) memcpy(c_last, cur, sizeof(float) * (size_t)ncols);
    free(cur);
    return 0;
}

/*
 * Reverse-time pass.
 * c        (T, B, d)  cell states saved by the forward
 * grad_h   (T, B, d)  grad_last (B, d) or NULL (zeros) = dL/dc_{T-1} from c_last
 * grad_u   (T, B, d, k)
 * grad_x   (T, B, d) when k == 3 (NULL when k == 4: the highway grad lands in grad_u[..,3])
 * grad_bias (2, B, d) per-(batch,column) partials (caller sums over B)
 * grad_c0  (B, d)
 */
int sru_oracle_bwd(const float *u, const float *x, const float *bias, const float *c0,
                   const float *c, const float *grad_h, const float *grad_last,
                   int T, int B, int d, int k, int act, float scale_x,
                   float *grad_u, float *grad_x, float *grad_bias, float *grad_c0)
{
    if (!(k == 3 || k == 4) || T < 0 || B <= 0 || d <= 0) return -1;
    if (k == 3 && (x == NULL || grad_x == NULL)) return -2;
    const ptrdiff_t ncols = (ptrdiff_t)B * d;

#pragma omp parallel for schedule(static)
    for (ptrdiff_t col = 0; col < ncols; ++col) {
        const int j = (int)(col % d);
        const float b1 = bias[j], b2 = bias[j + d];
        float gb1 = 0.0f, gb2 = 0.0f;
        float cur = grad_last ? grad_last[col] : 0.0f;
        for (int t = T - 1; t >= 0; --t) {
            const float *up = u + ((ptrdiff_t)t * ncols + col) * k;
            float *gup = grad_u + ((ptrdiff_t)t * ncols + col) * k;
            const float u0 = up[0];
            const float g1 = sigm(up[1] + b1);
            const float g2 = sigm(up[2] + b2);
            const float cval = act_fwd(act, c[(ptrdiff_t)t * ncols + col]);
            const float xv = (k == 3) ? x[(ptrdiff_t)t * ncols + col] * scale_x : up[3];
            const float cprev = (t > 0) ? c[(ptrdiff_t)(t - 1) * ncols + col]
                                        : (c0 ? c0[col] : 0.0f);
            const float gh = grad_h[(ptrdiff_t)t * ncols + col];

            /* h = (act(c) - x') * g2 + x' */
            const float gxp = gh * (1.0f - g2);
            if (k == 3) grad_x[(ptrdiff_t)t * ncols + col] = gxp * scale_x;
            else        gup[3] = gxp;
            const float gg2 = gh * (cval - xv) * (g2 * (1.0f - g2));
            gup[2] = gg2;
            gb2 += gg2;
            /* c = (c' - u0) * g1 + u0 */
            const float gc = gh * g2 * act_bwd(act, cval) + cur;
            gup[0] = gc * (1.0f - g1);
            const float gg1 = gc * (cprev - u0) * (g1 * (1.0f - g1));
            gup[1] = gg1;
            gb1 += gg1;
            cur = gc * g1;
        }
        grad_bias[col] = gb1;
        grad_bias[col + ncols] = gb2;
        grad_c0[col] = cur;
    }
    return 0;
}
