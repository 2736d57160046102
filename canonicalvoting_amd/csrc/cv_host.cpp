// Host-only pieces of libcvhip.so: error reporting and the per-class oriented-box
// NMS that follows the device decode (tens of boxes per scene; the reference runs
// it on the CPU too: eval_joint.py:75-89,270-280 with utils/calc_map.py:6-21).
#include "cv_common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

namespace {
thread_local char g_err[512] = "";

struct P2 { double x, y; };

double signed_area(const P2* p, int n) {
    double a = 0;
    for (int i = 0; i < n; ++i) {
        const P2& u = p[i];
        const P2& v = p[(i + 1) % n];
        a += u.x * v.y - v.x * u.y;
    }
    return 0.5 * a;
}

// Sutherland-Hodgman: keep the part of `in` on the left of the directed edge a->b.
int clip(const P2* in, int n, P2 a, P2 b, P2* out) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const P2 p = in[i], q = in[(i + 1) % n];
        const double sp = (b.x - a.x) * (p.y - a.y) - (b.y - a.y) * (p.x - a.x);
        const double sq = (b.x - a.x) * (q.y - a.y) - (b.y - a.y) * (q.x - a.x);
        if (sp >= 0) out[m++] = p;
        if ((sp > 0 && sq < 0) || (sp < 0 && sq > 0)) {
            const double t = sp / (sp - sq);
            out[m++] = P2{p.x + t * (q.x - p.x), p.y + t * (q.y - p.y)};
        }
    }
    return m;
}

double quad_overlap(const P2* q1, const P2* q2) {
    P2 a[4], b[4];
    std::memcpy(a, q1, sizeof a);
    std::memcpy(b, q2, sizeof b);
    if (signed_area(a, 4) < 0) std::swap(a[1], a[3]);
    if (signed_area(b, 4) < 0) std::swap(b[1], b[3]);
    P2 cur[16], nxt[16];
    int n = 4;
    std::memcpy(cur, a, sizeof a);
    for (int e = 0; e < 4 && n > 0; ++e) {
        n = clip(cur, n, b[e], b[(e + 1) % 4], nxt);
        std::memcpy(cur, nxt, sizeof(P2) * n);
    }
    return n < 3 ? 0.0 : std::fabs(signed_area(cur, n));
}
}  // namespace

void cv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" {

int cv_abi_version(void) { return CV_ABI_VERSION; }
const char* cv_last_error(void) { return g_err; }

// utils/calc_map.py:6-21.  Rows 0..3 of a box are its top face (the xz polygon),
// row 4 is on the bottom face; heights are float32 arithmetic (numpy scalars),
// areas are double (shapely).
double cv_iou_obb(const float* b1, const float* b2) {
    if (!b1 || !b2) return 0.0;
    if (!(b1[1] > b1[13] && b2[1] > b2[13])) return 0.0;   // :13
    P2 q1[4], q2[4];
    for (int i = 0; i < 4; ++i) {
        q1[i] = P2{b1[i * 3], b1[i * 3 + 2]};
        q2[i] = P2{b2[i * 3], b2[i * 3 + 2]};
    }
    const double inter_area = quad_overlap(q1, q2);
    const double a1 = std::fabs(signed_area(q1, 4)), a2 = std::fabs(signed_area(q2, 4));
    const float top = std::min(b1[1], b2[1]), bot = std::max(b1[13], b2[13]);
    volatile float ov = top - bot;
    const double inter_vol = inter_area * (ov > 0.0f ? (double)ov : 0.0);   // :18
    volatile float h1 = b1[1] - b1[13], h2 = b2[1] - b2[13];
    return inter_vol / (a1 * (double)h1 + a2 * (double)h2 - inter_vol);      // :19
}

// eval_joint.py:75-89: ascending (stable) argsort of the scores, repeatedly keep the
// last, drop everything overlapping it by more than `thr`.
int cv_nms_obb(const float* h_boxes, const float* h_scores, int n, double thr, int32_t* h_pick) {
    if (n < 0 || (n > 0 && (!h_boxes || !h_scores || !h_pick))) {
        cv_set_error("cv_nms_obb: bad arguments");
        return CV_EINVAL;
    }
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(),
                     [&](int a, int b) { return h_scores[a] < h_scores[b]; });
    int picked = 0;
    while (!order.empty()) {
        const int i = order.back();
        h_pick[picked++] = i;
        std::vector<int> keep;
        keep.reserve(order.size());
        for (size_t pos = 0; pos + 1 < order.size(); ++pos) {
            const int j = order[pos];
            if (!(cv_iou_obb(h_boxes + (size_t)i * 24, h_boxes + (size_t)j * 24) > thr))
                keep.push_back(j);
        }
        order.swap(keep);
    }
    return picked;
}

}  // extern "C"
