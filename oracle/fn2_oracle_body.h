/*
 * fn2_oracle_body.h -- type-generic body of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Included twice by fn2_oracle.c with T = float / double and SUF = f32 / f64.
 * Every function restates the arithmetic of one reference CUDA kernel of
 * NVIDIA/flownet2-pytorch (paths relative to the reference root), keeping
 *   - the type each product / accumulation is formed in,
 *   - the order partial sums are combined in (32-lane strided partials, then
 *     the shuffle tree / serial smem sum),
 * so that with -ffp-contract=off the result is what a literal, unfused
 * execution of the .cu source produces.  Nothing here is used by the product
 * path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may call it.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* ------------------------------------------------------------------------- */
/* correlation: channels_first  (correlation_cuda_kernel.cu:46-70)           */
/* rinput[n, y+pad, x+pad, c] = input[n, c, y, x]; rinput pre-zeroed          */
/* (correlation_cuda.cc:36-42).                                               */
static T *FN(o_channels_first)(const T *in, int B, int C, int H, int W, int pad)
{
    const long pH = H + 2L * pad, pW = W + 2L * pad;
    T *r = (T *)calloc((size_t)B * pH * pW * C, sizeof(T));
    if (!r) return NULL;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < B; ++n)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                T *dst = r + (((long)n * pH + (y + pad)) * pW + (x + pad)) * C;
                const T *src = in + (long)n * C * H * W + (long)y * W + x;
                for (int c = 0; c < C; ++c) dst[c] = src[(long)c * H * W];
            }
    return r;
}

/* padded-NHWC read; the reference reads out of bounds for kernel_size>1
 *  -- the oracle defines those reads as 0. */
static inline const T *FN(o_rvec)(const T *r, int n, long y, long x, long pH, long pW, int C)
{
    if (y < 0 || y >= pH || x < 0 || x >= pW) return NULL;
    return r + (((long)n * pH + y) * pW + x) * C;
}

/* correlation_forward  (correlation_cuda_kernel.cu:73-147)
 * + shape math of correlation_forward_cuda (correlation_cuda.cc:19-34). */
int FN(fn2o_corr_fwd)(const T *in1, const T *in2, T *out, int B, int C, int H, int W,
                      int pad, int k, int md, int s1, int s2)
{
    int pH, pW, nOut, oH, oW;
    if (fn2o_corr_shapes(H, W, pad, k, md, s1, s2, &pH, &pW, &nOut, &oH, &oW)) return -1;
    const int kr = (k - 1) / 2, dr = md / s2, D = 2 * dr + 1;
    const int nelems = k * k * C; /* int32_t nelems (:105) */
    T *r1 = FN(o_channels_first)(in1, B, C, H, W, pad);
    T *r2 = FN(o_channels_first)(in2, B, C, H, W, pad);
    if (!r1 || !r2) { free(r1); free(r2); return -2; }
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < B; ++n)
        for (int by = 0; by < oH; ++by)
            for (int bx = 0; bx < oW; ++bx) {
                const int y1 = by * s1 + md, x1 = bx * s1 + md; /* :90-91 */
                for (int tj = -dr; tj <= dr; ++tj)
                    for (int ti = -dr; ti <= dr; ++ti) {
                        const int x2 = x1 + ti * s2, y2 = y1 + tj * s2;
                        float lane[32]; /* float acc0 per lane (:112) */
                        for (int l = 0; l < 32; ++l) lane[l] = 0.0f;
                        for (int j = -kr; j <= kr; ++j)
                            for (int i = -kr; i <= kr; ++i) {
                                const T *a = FN(o_rvec)(r1, n, y1 + j, x1 + i, pH, pW, C);
                                const T *b = FN(o_rvec)(r2, n, y2 + j, x2 + i, pH, pW, C);
                                if (!a || !b) continue;
                                /* lane l strides channels by blockDim.x=32 (:118);
                                 * product in T, accumulated as float (:124) */
                                for (int ch = 0; ch < C; ++ch) {
                                    T p = a[ch] * b[ch];
                                    lane[ch & 31] += (float)p;
                                }
                            }
                        /* warpReduceSum: __shfl_down tree, offsets 16..1 (:16-21);
                         * only lane 0's value is stored. */
                        for (int off = 16; off > 0; off >>= 1)
                            for (int l = 0; l < off; ++l) lane[l] += lane[l + off];
                        const int tc = (tj + dr) * D + (ti + dr);
                        out[(((long)n * nOut + tc) * oH + by) * oW + bx] =
                            (T)(lane[0] / nelems); /* :143 */
                    }
            }
    free(r1);
    free(r2);
    return 0;
}

/* correlation_backward_input1 / _input2  (correlation_cuda_kernel.cu:150-241,
 * :243-334) launched per batch item on a (H, W, C) grid (:522-554).
 * Per output element: 32 partial sums prod_sum[tc % 32] accumulated in T over
 * tc = lane, lane+32, ...; lane 0 adds the 32 partials serially (:232-236),
 * then divides by nelems (T).  The loops below are re-nested (channel innermost)
 * but every element sees exactly that sequence of operations. */
int FN(fn2o_corr_bwd)(const T *in1, const T *in2, const T *gout, T *g1, T *g2,
                      int B, int C, int H, int W, int pad, int k, int md, int s1, int s2)
{
    int pH, pW, nOut, oH, oW;
    if (fn2o_corr_shapes(H, W, pad, k, md, s1, s2, &pH, &pW, &nOut, &oH, &oW)) return -1;
    const int kr = (k - 1) / 2, dr = md / s2, D = 2 * dr + 1;
    const T nelems = (T)(k * k * C); /* scalar_t nelems (:205) */
    T *r1 = FN(o_channels_first)(in1, B, C, H, W, pad);
    T *r2 = FN(o_channels_first)(in2, B, C, H, W, pad);
    if (!r1 || !r2) { free(r1); free(r2); return -2; }
    int fail = 0;
#pragma omp parallel
    {
        T *part = (T *)malloc(sizeof(T) * 32 * (size_t)C);
        if (!part) {
#pragma omp atomic write
            fail = 1;
        }
#pragma omp for collapse(2) schedule(static)
        for (int n = 0; n < B; ++n)
            for (int by = 0; by < H; ++by) {
                if (!part) continue;
                for (int bx = 0; bx < W; ++bx) {
                    const int y = by * s1 + pad, x = bx * s1 + pad; /* :161-162 */
                    /* ---------------- gradInput1 ---------------- */
                    {
                        int xmin = (x - kr - md) / s1, ymin = (y - kr - md) / s1; /* C trunc div (:171-175) */
                        int xmax = (x + kr - md) / s1, ymax = (y + kr - md) / s1;
                        int skip = (xmax < 0 || ymax < 0 || xmin >= oW || ymin >= oH) ||
                                   (xmin > xmax || ymin > ymax);
                        for (long q = 0; q < 32L * C; ++q) part[q] = 0;
                        if (!skip) {
                            if (xmin < 0) xmin = 0;
                            if (xmax > oW - 1) xmax = oW - 1;
                            if (ymin < 0) ymin = 0;
                            if (ymax > oH - 1) ymax = oH - 1;
                            for (int tc = 0; tc < nOut; ++tc) {
                                const int i2 = (tc % D - dr) * s2, j2 = (tc / D - dr) * s2;
                                const T *v2 = FN(o_rvec)(r2, n, y + j2, x + i2, pH, pW, C);
                                if (!v2) continue;
                                T *p = part + (long)(tc & 31) * C;
                                for (int j = ymin; j <= ymax; ++j)
                                    for (int i = xmin; i <= xmax; ++i) {
                                        const T go = gout[(((long)n * nOut + tc) * oH + j) * oW + i];
                                        for (int c = 0; c < C; ++c) p[c] += go * v2[c];
                                    }
                            }
                        }
                        /* the reference leaves skipped elements at their fill_(0) value */
                        for (int c = 0; c < C; ++c) {
                            T s = 0;
                            for (int l = 0; l < 32; ++l) s += part[(long)l * C + c];
                            g1[(((long)n * C + c) * H + by) * W + bx] = skip ? (T)0 : s / nelems;
                        }
                    }
                    /* ---------------- gradInput2 ---------------- */
                    {
                        for (long q = 0; q < 32L * C; ++q) part[q] = 0;
                        for (int tc = 0; tc < nOut; ++tc) {
                            const int i2 = (tc % D - dr) * s2, j2 = (tc / D - dr) * s2;
                            int xmin = (x - kr - md - i2) / s1, ymin = (y - kr - md - j2) / s1;
                            int xmax = (x + kr - md - i2) / s1, ymax = (y + kr - md - j2) / s1;
                            if (xmax < 0 || ymax < 0 || xmin >= oW || ymin >= oH) continue;
                            if (xmin > xmax || ymin > ymax) continue;
                            if (xmin < 0) xmin = 0;
                            if (xmax > oW - 1) xmax = oW - 1;
                            if (ymin < 0) ymin = 0;
                            if (ymax > oH - 1) ymax = oH - 1;
                            const T *v1 = FN(o_rvec)(r1, n, y - j2, x - i2, pH, pW, C);
                            if (!v1) continue;
                            T *p = part + (long)(tc & 31) * C;
                            for (int j = ymin; j <= ymax; ++j)
                                for (int i = xmin; i <= xmax; ++i) {
                                    const T go = gout[(((long)n * nOut + tc) * oH + j) * oW + i];
                                    for (int c = 0; c < C; ++c) p[c] += go * v1[c];
                                }
                        }
                        for (int c = 0; c < C; ++c) {
                            T s = 0;
                            for (int l = 0; l < 32; ++l) s += part[(long)l * C + c];
                            g2[(((long)n * C + c) * H + by) * W + bx] = s / nelems;
                        }
                    }
                }
            }
        free(part);
    }
    free(r1);
    free(r2);
    return fail ? -2 : 0;
}

/* ------------------------------------------------------------------------- */
/* channelnorm  (channelnorm_kernel.cu:18-60, :63-96)                         */
int FN(fn2o_chnorm_fwd)(const T *in, T *out, int B, int C, int H, int W)
{
    const long HW = (long)H * W;
#pragma omp parallel for schedule(static)
    for (long idx = 0; idx < (long)B * HW; ++idx) {
        const long b = idx / HW, p = idx % HW;
        float result = 0.0f; /* float result (:51) */
        for (int c = 0; c < C; ++c) {
            const T v = in[(b * C + c) * HW + p];
            result += (float)(v * v); /* square in T, accumulate float (:56) */
        }
        result = sqrtf(result); /* sqrt(float) (:58) */
        out[idx] = (T)result;
    }
    return 0;
}

/* gradOutput is read with its own batch stride (gs_b, in elements) so that the
 * oracle can serve both the reference's literal contiguous indexing (:92) and
 * the stride-honouring behaviour the replacement documents. */
int FN(fn2o_chnorm_bwd)(const T *in, const T *out, const T *gout, long gs_b, T *gin,
                        int B, int C, int H, int W)
{
    const long HW = (long)H * W;
#pragma omp parallel for schedule(static)
    for (long idx = 0; idx < (long)B * C * HW; ++idx) {
        const long b = idx / ((long)C * HW), p = idx % HW;
        /* float * float / (float + 1e-9 [double])  -> double divide -> float (:93) */
        float val = (float)((float)gout[b * gs_b + p] * (float)in[idx] /
                            ((float)out[b * HW + p] + 1e-9));
        gin[idx] = (T)val;
    }
    return 0;
}

#undef FN
#undef CAT
#undef CAT_
