/*
 * oracle/decode_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's heat-map decode for the ctdet and
 * multi_pose tasks.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this file.
 *
 * What it follows (paths relative to /root/reference/src/lib):
 *   _nms ........................ models/decode.py:9-15
 *   _topk_channel ............... models/decode.py:92-101
 *   _topk ....................... models/decode.py:103-119
 *   _gather_feat / _transpose_and_gather_feat ... models/utils.py:12-26
 *   ctdet_decode ................ models/decode.py:464-495
 *   multi_pose_decode ........... models/decode.py:497-571
 *
 * The arithmetic underneath (max_pool2d, topk, gather) lives in PyTorch, which
 * is a third-party dependency of the reference (readme/INSTALL.md:18-22 pins
 * 0.4.1; this image has 2.10).  Parity is pinned by running the reference's own
 * models/decode.py on CPU (tests/golden/gen_golden.py) and storing its outputs
 * under tests/golden/; tests/test_oracle_decode.py checks this file against them
 * bit for bit.
 *
 * Tie rule.  torch.topk leaves the order of equal scores unspecified.  This
 * restatement (and the HIP kernels) use one total order: score descending, then
 * class ascending, then spatial index ascending.  The golden vectors contain no
 * ties inside the top-K, so they cannot tell tie rules apart; tie cases are
 * tested oracle-vs-kernel only.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* models/decode.py:9-15 : 3x3 stride-1 max-pool with -inf padding, keep where
 * hmax == heat, output heat * keep (so suppressed cells become 0). */
void oracle_nms3x3(const float *heat, float *out, int planes, int H, int W)
{
    for (int p = 0; p < planes; ++p) {
        const float *src = heat + (size_t)p * H * W;
        float *dst = out + (size_t)p * H * W;
        for (int y = 0; y < H; ++y) {
            for (int x = 0; x < W; ++x) {
                float m = -INFINITY;
                for (int dy = -1; dy <= 1; ++dy) {
                    const int yy = y + dy;
                    if (yy < 0 || yy >= H)
                        continue;
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int xx = x + dx;
                        if (xx < 0 || xx >= W)
                            continue;
                        const float v = src[yy * W + xx];
                        if (v > m)
                            m = v;
                    }
                }
                const float v = src[y * W + x];
                const float keep = (m == v) ? 1.0f : 0.0f;
                dst[y * W + x] = v * keep;
            }
        }
    }
}

typedef struct {
    float score;
    int64_t idx; /* flat position inside the searched row */
} cand_t;

static int cand_cmp(const void *pa, const void *pb)
{
    const cand_t *a = (const cand_t *)pa, *b = (const cand_t *)pb;
    if (a->score > b->score)
        return -1;
    if (a->score < b->score)
        return 1;
    if (a->idx < b->idx)
        return -1;
    if (a->idx > b->idx)
        return 1;
    return 0;
}

/* top-K of one row of n floats: score desc, index asc.  n >= K required
 * (torch.topk raises otherwise). */
static int row_topk(const float *row, int64_t n, int K, float *out_score,
                    int64_t *out_idx)
{
    if (n < K)
        return -1;
    cand_t *c = (cand_t *)malloc(sizeof(cand_t) * (size_t)n);
    if (!c)
        return -2;
    for (int64_t i = 0; i < n; ++i) {
        c[i].score = row[i];
        c[i].idx = i;
    }
    qsort(c, (size_t)n, sizeof(cand_t), cand_cmp);
    for (int k = 0; k < K; ++k) {
        out_score[k] = c[k].score;
        out_idx[k] = c[k].idx;
    }
    free(c);
    return 0;
}

/* models/decode.py:92-101.  scores (B,C,H,W) -> (B,C,K) score / ind / ys / xs */
int oracle_topk_channel(const float *scores, int B, int C, int H, int W, int K,
                        float *topk_scores, int64_t *topk_inds, float *topk_ys,
                        float *topk_xs)
{
    const int64_t hw = (int64_t)H * W;
    int rc = 0;
#pragma omp parallel for schedule(dynamic)
    for (int bc = 0; bc < B * C; ++bc) {
        float *s = topk_scores + (size_t)bc * K;
        int64_t *ix = topk_inds + (size_t)bc * K;
        int r = row_topk(scores + (size_t)bc * hw, hw, K, s, ix);
        if (r != 0) {
            rc = r;
            continue;
        }
        for (int k = 0; k < K; ++k) {
            ix[k] = ix[k] % hw;
            topk_ys[(size_t)bc * K + k] = (float)(int)(ix[k] / W);
            topk_xs[(size_t)bc * K + k] = (float)(int)(ix[k] % W);
        }
    }
    return rc;
}

/* models/decode.py:103-119.  (B,C,H,W) -> (B,K) score / ind / cls / ys / xs */
int oracle_topk(const float *scores, int B, int C, int H, int W, int K,
                float *topk_score, int64_t *topk_inds, int32_t *topk_clses,
                float *topk_ys, float *topk_xs)
{
    const size_t n1 = (size_t)B * C * K;
    float *s1 = (float *)malloc(sizeof(float) * n1);
    int64_t *i1 = (int64_t *)malloc(sizeof(int64_t) * n1);
    float *y1 = (float *)malloc(sizeof(float) * n1);
    float *x1 = (float *)malloc(sizeof(float) * n1);
    int64_t *i2 = (int64_t *)malloc(sizeof(int64_t) * (size_t)K);
    int rc = -2;
    if (s1 && i1 && y1 && x1 && i2) {
        rc = oracle_topk_channel(scores, B, C, H, W, K, s1, i1, y1, x1);
        for (int b = 0; b < B && rc == 0; ++b) {
            /* second topk over the (C*K) per-class winners (decode.py:112) */
            rc = row_topk(s1 + (size_t)b * C * K, (int64_t)C * K, K,
                          topk_score + (size_t)b * K, i2);
            if (rc != 0)
                break;
            for (int k = 0; k < K; ++k) {
                const int64_t j = i2[k];
                topk_clses[(size_t)b * K + k] = (int32_t)(j / K); /* decode.py:113 */
                topk_inds[(size_t)b * K + k] = i1[(size_t)b * C * K + j];
                topk_ys[(size_t)b * K + k] = y1[(size_t)b * C * K + j];
                topk_xs[(size_t)b * K + k] = x1[(size_t)b * C * K + j];
            }
        }
    }
    free(s1);
    free(i1);
    free(y1);
    free(x1);
    free(i2);
    return rc;
}

/* models/utils.py:22-26: feat (B,Cf,H,W), ind (B,N) -> out (B,N,Cf) */
void oracle_transpose_and_gather_feat(const float *feat, const int64_t *ind,
                                      int B, int Cf, int H, int W, int N,
                                      float *out)
{
    const size_t hw = (size_t)H * W;
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n)
            for (int c = 0; c < Cf; ++c)
                out[((size_t)b * N + n) * Cf + c] =
                    feat[((size_t)b * Cf + c) * hw + (size_t)ind[(size_t)b * N + n]];
}

/*
 * models/decode.py:464-495.
 *   heat (B,C,H,W) POST-sigmoid; wh (B,2,H,W) or (B,2C,H,W) if cat_spec_wh;
 *   reg (B,2,H,W) or NULL.
 *   dets (B,K,6) = [x1,y1,x2,y2,score,cls]; inds_out (B,K) optional.
 */
int oracle_ctdet_decode(const float *heat, const float *wh, const float *reg,
                        int B, int C, int H, int W, int K, int cat_spec_wh,
                        float *dets, int64_t *inds_out)
{
    const size_t n = (size_t)B * C * H * W;
    float *nmsd = (float *)malloc(sizeof(float) * n);
    float *score = (float *)malloc(sizeof(float) * (size_t)B * K);
    int64_t *inds = (int64_t *)malloc(sizeof(int64_t) * (size_t)B * K);
    int32_t *cls = (int32_t *)malloc(sizeof(int32_t) * (size_t)B * K);
    float *ys = (float *)malloc(sizeof(float) * (size_t)B * K);
    float *xs = (float *)malloc(sizeof(float) * (size_t)B * K);
    int rc = -2;
    if (nmsd && score && inds && cls && ys && xs) {
        oracle_nms3x3(heat, nmsd, B * C, H, W);
        rc = oracle_topk(nmsd, B, C, H, W, K, score, inds, cls, ys, xs);
    }
    if (rc == 0) {
        const size_t hw = (size_t)H * W;
        const int whC = cat_spec_wh ? 2 * C : 2;
        for (int b = 0; b < B; ++b) {
            for (int k = 0; k < K; ++k) {
                const size_t bk = (size_t)b * K + k;
                const size_t ind = (size_t)inds[bk];
                float x = xs[bk], y = ys[bk];
                if (reg) { /* decode.py:472-476 */
                    x = x + reg[((size_t)b * 2 + 0) * hw + ind];
                    y = y + reg[((size_t)b * 2 + 1) * hw + ind];
                } else { /* decode.py:477-479 */
                    x = x + 0.5f;
                    y = y + 0.5f;
                }
                float w, h;
                if (cat_spec_wh) { /* decode.py:481-484 */
                    const int c = cls[bk];
                    w = wh[((size_t)b * whC + 2 * c + 0) * hw + ind];
                    h = wh[((size_t)b * whC + 2 * c + 1) * hw + ind];
                } else {
                    w = wh[((size_t)b * whC + 0) * hw + ind];
                    h = wh[((size_t)b * whC + 1) * hw + ind];
                }
                float *d = dets + bk * 6;
                d[0] = x - w / 2; /* decode.py:489-492 */
                d[1] = y - h / 2;
                d[2] = x + w / 2;
                d[3] = y + h / 2;
                d[4] = score[bk];
                d[5] = (float)cls[bk];
                if (inds_out)
                    inds_out[bk] = inds[bk];
            }
        }
    }
    free(nmsd);
    free(score);
    free(inds);
    free(cls);
    free(ys);
    free(xs);
    return rc;
}

/*
 * models/decode.py:497-571.
 *   heat (B,C,H,W) post-sigmoid, wh (B,2,H,W), kps (B,2J,H,W), reg (B,2,H,W)|NULL,
 *   hm_hp (B,J,H,W) post-sigmoid |NULL, hp_offset (B,2,H,W)|NULL.
 *   dets (B,K,4+1+2J+1).
 */
int oracle_multi_pose_decode(const float *heat, const float *wh,
                             const float *kps_map, const float *reg,
                             const float *hm_hp, const float *hp_offset, int B,
                             int C, int H, int W, int J, int K, float *dets)
{
    const size_t hw = (size_t)H * W;
    const size_t n = (size_t)B * C * hw;
    const int D = 4 + 1 + 2 * J + 1;
    float *nmsd = (float *)malloc(sizeof(float) * n);
    float *score = (float *)malloc(sizeof(float) * (size_t)B * K);
    int64_t *inds = (int64_t *)malloc(sizeof(int64_t) * (size_t)B * K);
    int32_t *cls = (int32_t *)malloc(sizeof(int32_t) * (size_t)B * K);
    float *ys = (float *)malloc(sizeof(float) * (size_t)B * K);
    float *xs = (float *)malloc(sizeof(float) * (size_t)B * K);
    float *kps = (float *)malloc(sizeof(float) * (size_t)B * K * 2 * J);
    float *bbox = (float *)malloc(sizeof(float) * (size_t)B * K * 4);
    int rc = -2;
    if (nmsd && score && inds && cls && ys && xs && kps && bbox) {
        oracle_nms3x3(heat, nmsd, B * C, H, W);
        rc = oracle_topk(nmsd, B, C, H, W, K, score, inds, cls, ys, xs);
    }
    if (rc == 0) {
        for (int b = 0; b < B; ++b) {
            for (int k = 0; k < K; ++k) {
                const size_t bk = (size_t)b * K + k;
                const size_t ind = (size_t)inds[bk];
                /* decode.py:506-509 : kps += (xs, ys) BEFORE reg is added */
                for (int j = 0; j < J; ++j) {
                    kps[bk * 2 * J + 2 * j] =
                        kps_map[((size_t)b * 2 * J + 2 * j) * hw + ind] + xs[bk];
                    kps[bk * 2 * J + 2 * j + 1] =
                        kps_map[((size_t)b * 2 * J + 2 * j + 1) * hw + ind] + ys[bk];
                }
                float x = xs[bk], y = ys[bk];
                if (reg) {
                    x = x + reg[((size_t)b * 2 + 0) * hw + ind];
                    y = y + reg[((size_t)b * 2 + 1) * hw + ind];
                } else {
                    x = x + 0.5f;
                    y = y + 0.5f;
                }
                const float w = wh[((size_t)b * 2 + 0) * hw + ind];
                const float h = wh[((size_t)b * 2 + 1) * hw + ind];
                bbox[bk * 4 + 0] = x - w / 2;
                bbox[bk * 4 + 1] = y - h / 2;
                bbox[bk * 4 + 2] = x + w / 2;
                bbox[bk * 4 + 3] = y + h / 2;
            }
        }
    }
    if (rc == 0 && hm_hp) { /* decode.py:527-568 */
        const float thresh = 0.1f;
        const size_t nj = (size_t)B * J * hw;
        float *hp_n = (float *)malloc(sizeof(float) * nj);
        float *hs = (float *)malloc(sizeof(float) * (size_t)B * J * K);
        int64_t *hi = (int64_t *)malloc(sizeof(int64_t) * (size_t)B * J * K);
        float *hy = (float *)malloc(sizeof(float) * (size_t)B * J * K);
        float *hx = (float *)malloc(sizeof(float) * (size_t)B * J * K);
        if (!(hp_n && hs && hi && hy && hx)) {
            rc = -2;
        } else {
            oracle_nms3x3(hm_hp, hp_n, B * J, H, W);
            rc = oracle_topk_channel(hp_n, B, J, H, W, K, hs, hi, hy, hx);
        }
        if (rc == 0) {
            for (int b = 0; b < B; ++b) {
                for (int j = 0; j < J; ++j) {
                    float *s = hs + ((size_t)b * J + j) * K;
                    float *py = hy + ((size_t)b * J + j) * K;
                    float *px = hx + ((size_t)b * J + j) * K;
                    const int64_t *pi = hi + ((size_t)b * J + j) * K;
                    for (int k = 0; k < K; ++k) {
                        if (hp_offset) { /* decode.py:534-539 */
                            px[k] = px[k] + hp_offset[((size_t)b * 2 + 0) * hw + (size_t)pi[k]];
                            py[k] = py[k] + hp_offset[((size_t)b * 2 + 1) * hw + (size_t)pi[k]];
                        } else {
                            px[k] = px[k] + 0.5f;
                            py[k] = py[k] + 0.5f;
                        }
                        /* decode.py:544-547 */
                        const float m = (s[k] > thresh) ? 1.0f : 0.0f;
                        s[k] = (1 - m) * -1 + m * s[k];
                        py[k] = (1 - m) * (-10000) + m * py[k];
                        px[k] = (1 - m) * (-10000) + m * px[k];
                    }
                    for (int k = 0; k < K; ++k) { /* detection k */
                        const size_t bk = (size_t)b * K + k;
                        const float rx = kps[bk * 2 * J + 2 * j];
                        const float ry = kps[bk * 2 * J + 2 * j + 1];
                        /* decode.py:550-551 : dist over candidates, first minimum */
                        float best = 0.f;
                        int bi = -1;
                        for (int q = 0; q < K; ++q) {
                            const float dx = rx - px[q];
                            const float dy = ry - py[q];
                            const float d = sqrtf(dx * dx + dy * dy);
                            if (bi < 0 || d < best) {
                                best = d;
                                bi = q;
                            }
                        }
                        const float sc = s[bi];
                        const float kx = px[bi], ky = py[bi];
                        const float l = bbox[bk * 4 + 0], t = bbox[bk * 4 + 1];
                        const float r = bbox[bk * 4 + 2], bt = bbox[bk * 4 + 3];
                        const float bh = bt - t, bw = r - l;
                        const float mx = (bh > bw ? bh : bw) * 0.3f;
                        /* decode.py:562-565 */
                        const int reject = (kx < l) || (kx > r) || (ky < t) ||
                                           (ky > bt) || (sc < thresh) || (best > mx);
                        const float m = reject ? 1.0f : 0.0f;
                        /* decode.py:566 */
                        kps[bk * 2 * J + 2 * j] = (1 - m) * kx + m * rx;
                        kps[bk * 2 * J + 2 * j + 1] = (1 - m) * ky + m * ry;
                    }
                }
            }
        }
        free(hp_n);
        free(hs);
        free(hi);
        free(hy);
        free(hx);
    }
    if (rc == 0) {
        for (size_t bk = 0; bk < (size_t)B * K; ++bk) {
            float *d = dets + bk * D;
            memcpy(d, bbox + bk * 4, sizeof(float) * 4);
            d[4] = score[bk];
            memcpy(d + 5, kps + bk * 2 * J, sizeof(float) * 2 * J);
            d[5 + 2 * J] = (float)cls[bk];
        }
    }
    free(nmsd);
    free(score);
    free(inds);
    free(cls);
    free(ys);
    free(xs);
    free(kps);
    free(bbox);
    return rc;
}
