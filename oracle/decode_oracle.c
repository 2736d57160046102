/*
 * oracle/decode_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Sequential CPU restatement of the reference's inline detection decode
 * (greedy peak picking, grid suppression, LCC-aware back-projection check,
 * class vote, per-class oriented-box NMS).  Only tests/, smoke() and
 * bench.py's cpu_baseline leg may use it.
 *
 * Follows:
 *   cv_oracle_decode  <- eval_joint.py:195-263 (canonical copy; constants :18-21,
 *                        unravel_index :41-46); `elim_hi_plus1 = 0` gives the
 *                        eval_separate.py:209 variant of the elimination slice.
 *   cv_oracle_iou_obb <- utils/calc_map.py:6-21 (shapely polygons restated as
 *                        convex-quad clipping in double)
 *   cv_oracle_nms     <- eval_joint.py:75-89, applied per class as :270-280
 *
 * PARITY STATUS: cv_oracle_decode is PINNED BY REFERENCE EXECUTION: the reference
 * loop is inline in main() (not importable), so tests/golden/make_decode_golden.py
 * slices eval_joint.py:195-263 out of the file and exec()s it on CPU torch in the
 * build container; tests/test_oracle_decode.py checks this file against the
 * resulting tests/golden/decode_ref_*.npz (box count, classes, zeroed cells exact;
 * boxes <= 2e-6), next to the hand-built known-answer grids.  The IoU/NMS pair is
 * pinned by tests/golden/map_golden.npz and analytic values (shapely is absent).
 *
 * fp32 conventions where torch-on-GPU leaves the rounding implementation
 * defined (all shared with the HIP decode so both agree bit for bit):
 *   - atan2/cos/sin are evaluated in double and rounded to float.
 *   - [M,3]@[3,3] products are a k-ordered fmaf chain (what a GEMM does).
 *   - tensor / python_scalar is tensor * (1.f/scalar) (torch CUDA kernels'
 *     CPU-scalar fast path), tensor / tensor is a true division.
 *   - the masked mean of eval_joint.py:250 is summed in double, divided by the
 *     count and rounded to float (the GPU tree-reduction order is unspecified).
 *   - NMS sorts by (score, index) ascending, i.e. a stable argsort.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float thresh_high;   /* eval_joint.py:18  = 60  */
    float thresh_low;    /* :19 = 10 */
    float valid_ratio;   /* :20 = 0.2 */
    int elimination;     /* :21 = 2 */
    float prob_thresh;   /* :245 = 0.3 */
    int elim_hi_plus1;   /* 1: eval_joint.py:211 (c+e+1); 0: eval_separate.py:209 (c+e) */
    int max_iters;       /* safety bound on the while-True loop */
    double err_thresh;   /* :252 = 0.3 (python float compared with .item()) */
} cv_decode_params;

/* verdict codes per examined candidate */
enum { CV_ACCEPT = 0, CV_REJ_FEW = 1, CV_REJ_ERR = 2 };

static const float RAWX[8] = {1, 1, -1, -1, 1, 1, -1, -1};
static const float RAWY[8] = {1, 1, 1, 1, -1, -1, -1, -1};
static const float RAWZ[8] = {1, -1, -1, 1, 1, -1, -1, 1};

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Returns the number of candidates examined (<= max_iters); n_boxes_out gets the
 * number of accepted boxes.  grid_obj is mutated in place like the reference. */
int cv_oracle_decode(float* grid_obj, const float* grid_rot, const float* grid_scale,
                     const int dims[3], const float corner[3], float res,
                     const float* points, const float* xyz_pred, const float* prob_pred,
                     const int32_t* class_pred, int64_t n, const cv_decode_params* p,
                     int64_t* cand_idx /*[max_iters]*/, int32_t* verdict /*[max_iters]*/,
                     float* boxes /*[max_iters][8][3]*/, float* scores /*[max_iters]*/,
                     int32_t* classes /*[max_iters]*/, int* n_boxes_out) {
    const int X = dims[0], Y = dims[1], Z = dims[2];
    const int64_t G = (int64_t)X * Y * Z;
    const float inv_res = 1.0f / res;
    int n_boxes = 0, it = 0;
    for (; it < p->max_iters; ++it) {
        /* :205 first maximum in flat order */
        int64_t best = 0;
        float bv = grid_obj[0];
        for (int64_t i = 1; i < G; ++i)
            if (grid_obj[i] > bv) { bv = grid_obj[i]; best = i; }
        const int cz = (int)(best % Z), cy = (int)((best / Z) % Y), cx = (int)(best / ((int64_t)Z * Y));
        const int cand[3] = {cx, cy, cz};
        /* :206 */
        float cw[3];
        for (int k = 0; k < 3; ++k) {
            volatile float t = res * (float)cand[k];
            cw[k] = corner[k] + t;
        }
        if (bv < p->thresh_high) break; /* :208-209 */
        cand_idx[it] = best;
        /* :211 */
        {
            const int e = p->elimination, hp = e + (p->elim_hi_plus1 ? 1 : 0);
            const int x0 = cx - e < 0 ? 0 : cx - e, y0 = cy - e < 0 ? 0 : cy - e,
                      z0 = cz - e < 0 ? 0 : cz - e;
            const int x1 = cx + hp > X ? X : cx + hp, y1 = cy + hp > Y ? Y : cy + hp,
                      z1 = cz + hp > Z ? Z : cz + hp;
            for (int x = x0; x < x1; ++x)
                for (int y = y0; y < y1; ++y)
                    for (int z = z0; z < z1; ++z) grid_obj[((int64_t)x * Y + y) * Z + z] = 0.f;
        }
        /* :213-216 */
        const float r0 = grid_rot[best * 2 + 0], r1 = grid_rot[best * 2 + 1];
        const float rot = (float)atan2((double)r1, (double)r0);
        const float c = (float)cos((double)rot), s = (float)sin((double)rot);
        const float sc[3] = {grid_scale[best * 3 + 0], grid_scale[best * 3 + 1],
                             grid_scale[best * 3 + 2]};
        /* R = [[c,0,-s],[0,1,0],[s,0,c]];  M = R @ diag(sc) */
        const float m00 = c * sc[0], m02 = (-s) * sc[2], m11 = sc[1], m20 = s * sc[0],
                    m22 = c * sc[2];
        float bb[8][3];
        float lo[3], hi[3];
        for (int q = 0; q < 8; ++q) {
            bb[q][0] = m00 * RAWX[q] + m02 * RAWZ[q];
            bb[q][1] = m11 * RAWY[q];
            bb[q][2] = m20 * RAWX[q] + m22 * RAWZ[q];
            for (int k = 0; k < 3; ++k) {
                if (q == 0 || bb[q][k] < lo[k]) lo[k] = bb[q][k];
                if (q == 0 || bb[q][k] > hi[k]) hi[k] = bb[q][k];
            }
        }
        /* :220-223 */
        int blo[3], bhi[3], clo[3], chi[3];
        const int shape[3] = {X, Y, Z};
        for (int k = 0; k < 3; ++k) {
            volatile float a = lo[k] * inv_res, b = hi[k] * inv_res;
            blo[k] = (int)a;
            bhi[k] = (int)b;
            clo[k] = clampi(cand[k] + blo[k], 0, shape[k] - 1);
            chi[k] = clampi(cand[k] + bhi[k], 0, shape[k] - 1);
        }
        /* :225-229, :243 */
        for (int x = clo[0]; x <= chi[0]; ++x)
            for (int y = clo[1]; y <= chi[1]; ++y)
                for (int z = clo[2]; z <= chi[2]; ++z) {
                    volatile float v0 = (float)(x - cx) * res, v1 = (float)(y - cy) * res,
                                   v2 = (float)(z - cz) * res;
                    const float i0 = fmaf(v2, s, fmaf(v1, 0.f, v0 * c)) / sc[0];
                    const float i1 = fmaf(v2, 0.f, fmaf(v1, 1.f, v0 * 0.f)) / sc[1];
                    const float i2 = fmaf(v2, c, fmaf(v1, 0.f, v0 * (-s))) / sc[2];
                    if (-1 < i0 && i0 < 1 && -1 < i1 && i1 < 1 && -1 < i2 && i2 < 1)
                        grid_obj[((int64_t)x * Y + y) * Z + z] = 0.f;
                }
        /* :231-234, :245-250 */
        int64_t n_in = 0, n_mask = 0;
        double err_sum = 0.0;
        float probmax = -INFINITY;
        int64_t hist[64];
        memset(hist, 0, sizeof hist);
        for (int64_t i = 0; i < n; ++i) {
            volatile float d0 = points[i * 3 + 0] - cw[0], d1 = points[i * 3 + 1] - cw[1],
                           d2 = points[i * 3 + 2] - cw[2];
            const float w0 = fmaf(d2, s, fmaf(d1, 0.f, d0 * c)) / sc[0];
            const float w1 = fmaf(d2, 0.f, fmaf(d1, 1.f, d0 * 0.f)) / sc[1];
            const float w2 = fmaf(d2, c, fmaf(d1, 0.f, d0 * (-s))) / sc[2];
            if (!(-1 < w0 && w0 < 1 && -1 < w1 && w1 < 1 && -1 < w2 && w2 < 1)) continue;
            ++n_in;
            const float pr = prob_pred[i];
            if (pr > probmax) probmax = pr;
            if (!(pr > p->prob_thresh)) continue;
            ++n_mask;
            volatile float e0 = xyz_pred[i * 3 + 0] - w0, e1 = xyz_pred[i * 3 + 1] - w1,
                           e2 = xyz_pred[i * 3 + 2] - w2;
            volatile float q0 = e0 * e0, q1 = e1 * e1, q2 = e2 * e2;
            volatile float ss = (q0 + q1) + q2;
            volatile float nr = sqrtf(ss);
            volatile float term = nr * pr;
            err_sum += (double)term;
            const int cl = class_pred[i];
            if (cl >= 0 && cl < 64) ++hist[cl];
        }
        /* :246-247 */
        {
            volatile float lhs = (float)n_mask, rhs = p->valid_ratio * (float)n_in;
            if (lhs < rhs || (float)n_in < p->thresh_low) { verdict[it] = CV_REJ_FEW; continue; }
        }
        /* :249-253 */
        const float error = (float)(err_sum / (double)n_mask);
        if ((double)error > p->err_thresh) { verdict[it] = CV_REJ_ERR; continue; }
        /* :255-256 smallest class id among the most frequent */
        int best_cls = 0;
        int64_t best_cnt = -1;
        for (int k = 0; k < 64; ++k)
            if (hist[k] > best_cnt) { best_cnt = hist[k]; best_cls = k; }
        /* :258-263 */
        for (int q = 0; q < 8; ++q)
            for (int k = 0; k < 3; ++k) boxes[((int64_t)n_boxes * 8 + q) * 3 + k] = bb[q][k] + cw[k];
        scores[n_boxes] = probmax;
        classes[n_boxes] = best_cls;
        verdict[it] = CV_ACCEPT;
        ++n_boxes;
    }
    *n_boxes_out = n_boxes;
    return it;
}

/* ---- utils/calc_map.py:6-21 ------------------------------------------------ */
typedef struct { double x, y; } pt2;

static double poly_area(const pt2* p, int n) {
    double a = 0;
    for (int i = 0; i < n; ++i) {
        const pt2 u = p[i], v = p[(i + 1) % n];
        a += u.x * v.y - v.x * u.y;
    }
    return 0.5 * a;
}

/* clip subject polygon by the half plane left of a->b (clip polygon CCW) */
static int clip_edge(const pt2* in, int n, pt2 a, pt2 b, pt2* out) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const pt2 P = in[i], Q = in[(i + 1) % n];
        const double sp = (b.x - a.x) * (P.y - a.y) - (b.y - a.y) * (P.x - a.x);
        const double sq = (b.x - a.x) * (Q.y - a.y) - (b.y - a.y) * (Q.x - a.x);
        if (sp >= 0) out[m++] = P;
        if ((sp > 0 && sq < 0) || (sp < 0 && sq > 0)) {
            const double t = sp / (sp - sq);
            pt2 I = {P.x + t * (Q.x - P.x), P.y + t * (Q.y - P.y)};
            out[m++] = I;
        }
    }
    return m;
}

static double quad_intersection_area(const pt2* q1, const pt2* q2) {
    pt2 a[4], b[4];
    memcpy(a, q1, sizeof a);
    memcpy(b, q2, sizeof b);
    if (poly_area(a, 4) < 0) { pt2 t = a[1]; a[1] = a[3]; a[3] = t; }
    if (poly_area(b, 4) < 0) { pt2 t = b[1]; b[1] = b[3]; b[3] = t; }
    pt2 buf1[16], buf2[16];
    int n = 4;
    memcpy(buf1, a, sizeof a);
    for (int e = 0; e < 4 && n > 0; ++e) {
        n = clip_edge(buf1, n, b[e], b[(e + 1) % 4], buf2);
        memcpy(buf1, buf2, sizeof(pt2) * n);
    }
    if (n < 3) return 0.0;
    return fabs(poly_area(buf1, n));
}

double cv_oracle_iou_obb(const float* b1 /*[8][3]*/, const float* b2) {
    /* :13 */
    if (!(b1[0 * 3 + 1] > b1[4 * 3 + 1] && b2[0 * 3 + 1] > b2[4 * 3 + 1])) return 0.0;
    pt2 q1[4], q2[4];
    for (int i = 0; i < 4; ++i) {
        q1[i].x = b1[i * 3 + 0]; q1[i].y = b1[i * 3 + 2];
        q2[i].x = b2[i * 3 + 0]; q2[i].y = b2[i * 3 + 2];
    }
    const double inter_area = quad_intersection_area(q1, q2);
    const double a1 = fabs(poly_area(q1, 4)), a2 = fabs(poly_area(q2, 4));
    /* :18 heights are numpy float32 arithmetic */
    const float top = b1[1] < b2[1] ? b1[1] : b2[1];
    const float bot = b1[4 * 3 + 1] > b2[4 * 3 + 1] ? b1[4 * 3 + 1] : b2[4 * 3 + 1];
    volatile float ov = top - bot;
    const double h = ov > 0.0f ? (double)ov : 0.0;
    const double inter_vol = inter_area * h;
    volatile float h1 = b1[1] - b1[4 * 3 + 1], h2 = b2[1] - b2[4 * 3 + 1];
    return inter_vol / (a1 * (double)h1 + a2 * (double)h2 - inter_vol);
}

/* eval_joint.py:75-89 on one class' boxes; returns the pick count. */
int cv_oracle_nms(const float* boxes, const float* scores, int n, double thr, int32_t* pick) {
    int* I = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) I[i] = i;
    /* stable ascending argsort (insertion sort; n is tens) */
    for (int i = 1; i < n; ++i) {
        int v = I[i], j = i - 1;
        while (j >= 0 && scores[I[j]] > scores[v]) { I[j + 1] = I[j]; --j; }
        I[j + 1] = v;
    }
    int m = n, np_ = 0;
    while (m > 0) {
        const int i = I[m - 1];
        pick[np_++] = i;
        int w = 0;
        for (int pos = 0; pos < m - 1; ++pos) {
            const int j = I[pos];
            const double o = cv_oracle_iou_obb(boxes + (int64_t)i * 24, boxes + (int64_t)j * 24);
            if (!(o > thr)) I[w++] = j;
        }
        m = w;
    }
    free(I);
    return np_;
}
