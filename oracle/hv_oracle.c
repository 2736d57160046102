/*
 * oracle/hv_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement (strict IEEE fp32, no FMA contraction, no fast-math)
 * of the reference canonical-vote op.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (canonicalvoting_amd/) never does.
 *
 * Follows, statement by statement:
 *   hv_oracle_grid_dims   <- houghvoting/src/hv_cuda_kernel.cu:129-134 (V1)
 *   hv_oracle_forward     <- houghvoting/src/hv_cuda_kernel.cu:25-96   (V2)
 *   hv_oracle_average     <- houghvoting/src/hv_cuda_kernel.cu:106-118 (V3)
 *   hv_oracle_backward    <- houghvoting/src/hv_cuda_kernel.cu:183-260 (V4)
 *   helper semantics      <- houghvoting/src/helper_math.h:157-160 (int3 trunc),
 *                            :994-997 (float3/float true division),
 *                            :1338-1341 (fracf = v - floorf(v))
 *
 * PARITY STATUS: "parity unpinned".  The reference ships no tests, golden
 * vectors or CPU path for this op (hv_cuda.cpp:26-28 rejects CPU tensors) and
 * its .cu cannot be built here (needs nvcc + torch CUDA headers), so this
 * oracle is pinned only by (a) an independent numpy restatement
 * (oracle/hv_numpy.py) and (b) analytic known-answer tests in tests/.
 *
 * Two deliberate, documented conventions (the CUDA binary cannot be bit-matched
 * on either, and both are far below the 1e-4 float tolerance):
 *   - cos/sin of theta are taken as (float)cos((double)theta): the correctly
 *     rounded value.  The reference calls cosf/sinf on device (<=2 ulp).
 *   - multiply-adds are NOT fused (nvcc would contract some of them).
 * The product HIP kernel uses the same two conventions, so cell indices and
 * per-vote contributions are bit-identical; only the fp32 summation order of
 * the atomics differs (as it does run-to-run in the reference itself).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define IDX3(x, y, z, Y, Z) ((((int64_t)(x)) * (Y) + (y)) * (Z) + (z))

/* hv_cuda_kernel.cu:129-134: corners = stack(min(points,0), max(points,0));
 * diff = (corners[1]-corners[0]) / res  (fp32 tensor ops);
 * size_k = diff[k].item().to<int>() + 1 (truncation). */
void hv_oracle_minmax(const float* pts, int64_t n, float mn[3], float mx[3]) {
    for (int k = 0; k < 3; ++k) { mn[k] = pts[k]; mx[k] = pts[k]; }
    for (int64_t i = 1; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            float v = pts[i * 3 + k];
            if (v < mn[k]) mn[k] = v;
            if (v > mx[k]) mx[k] = v;
        }
}

void hv_oracle_grid_dims(const float mn[3], const float mx[3], float res, int dims[3]) {
    for (int k = 0; k < 3; ++k) {
        volatile float d = mx[k] - mn[k];
        volatile float q = d / res;
        dims[k] = (int)q + 1;
    }
}

/* Table of (cos(theta_i), sin(theta_i)), theta_i = i * (2*3.141592654f/num_rots)
 * computed in fp32 exactly as hv_cuda_kernel.cu:35,37. */
void hv_oracle_rot_table(int num_rots, float* cs /* [num_rots][2] */) {
    const float rot_interval = 2 * 3.141592654f / num_rots;
    for (int i = 0; i < num_rots; ++i) {
        float theta = i * rot_interval;
        cs[2 * i + 0] = (float)cos((double)theta);
        cs[2 * i + 1] = (float)sin((double)theta);
    }
}

/* hv_cuda_kernel.cu:25-96.  Grids must be zero-filled by the caller
 * (torch::zeros at :132-134).  Accumulation order: point-major, rot-minor,
 * i.e. the order a single serial thread would execute the reference loop.
 * Returns the number of in-bounds votes (V_in of SURVEY 8d). */
int64_t hv_oracle_forward(const float* points, const float* xyz, const float* scale,
                          const float* obj, int64_t n, float res, int num_rots,
                          const float corner[3], const int dims[3],
                          float* g_obj, float* g_rot, float* g_scale) {
    const int X = dims[0], Y = dims[1], Z = dims[2];
    float cs[2 * 4096];
    if (num_rots > 4096) return -1;
    hv_oracle_rot_table(num_rots, cs);
    int64_t v_in = 0;
    for (int64_t c = 0; c < n; ++c) {
        const float objness = obj[c];
        const float cx = xyz[c * 3 + 0] * scale[c * 3 + 0];
        const float cy = xyz[c * 3 + 1] * scale[c * 3 + 1];
        const float cz = xyz[c * 3 + 2] * scale[c * 3 + 2];
        const float px = points[c * 3 + 0], py = points[c * 3 + 1], pz = points[c * 3 + 2];
        for (int i = 0; i < num_rots; ++i) {
            const float ct = cs[2 * i], st = cs[2 * i + 1];
            /* :38-39 */
            const float ox = (-ct) * cx + st * cz;
            const float oy = -cy;
            const float oz = (-st) * cx - ct * cz;
            /* :40 (point + offset - corner) / res */
            const float gx = ((px + ox) - corner[0]) / res;
            const float gy = ((py + oy) - corner[1]) / res;
            const float gz = ((pz + oz) - corner[2]) / res;
            /* :41-44 */
            if (gx < 0 || gy < 0 || gz < 0 || gx >= (float)(X - 1) || gy >= (float)(Y - 1) ||
                gz >= (float)(Z - 1))
                continue;
            ++v_in;
            const int fx = (int)gx, fy = (int)gy, fz = (int)gz; /* :45 */
            const int hx = fx + 1, hy = fy + 1, hz = fz + 1;    /* :46 */
            const float rx = gx - floorf(gx), ry = gy - floorf(gy), rz = gz - floorf(gz); /* :47 */
            const float w0x = 1.f - rx, w0y = 1.f - ry, w0z = 1.f - rz; /* :49 */
            const float w1x = rx, w1y = ry, w1z = rz;                    /* :50 */
            /* :52-59 */
            const float w[8] = {
                w0x * w0y * w0z * objness, w0x * w0y * w1z * objness,
                w0x * w1y * w0z * objness, w0x * w1y * w1z * objness,
                w1x * w0y * w0z * objness, w1x * w0y * w1z * objness,
                w1x * w1y * w0z * objness, w1x * w1y * w1z * objness};
            const int64_t cell[8] = {
                IDX3(fx, fy, fz, Y, Z), IDX3(fx, fy, hz, Y, Z), IDX3(fx, hy, fz, Y, Z),
                IDX3(fx, hy, hz, Y, Z), IDX3(hx, fy, fz, Y, Z), IDX3(hx, fy, hz, Y, Z),
                IDX3(hx, hy, fz, Y, Z), IDX3(hx, hy, hz, Y, Z)};
            for (int k = 0; k < 8; ++k) g_obj[cell[k]] += w[k]; /* :61-68 */
            const float rot_vec[2] = {ct, st};                  /* :70 */
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 8; ++k) g_rot[cell[k] * 2 + j] += w[k] * rot_vec[j];
            for (int j = 0; j < 3; ++j) { /* :83-93 */
                const float s = scale[c * 3 + j];
                for (int k = 0; k < 8; ++k) g_scale[cell[k] * 3 + j] += w[k] * s;
            }
        }
    }
    return v_in;
}

/* hv_cuda_kernel.cu:106-118: `x /= w + 1e-7` with a double literal: the sum
 * and the quotient are evaluated in double and rounded to float on store. */
void hv_oracle_average(const float* g_obj, float* g_rot, float* g_scale, int64_t cells) {
    for (int64_t i = 0; i < cells; ++i) {
        const double d = (double)g_obj[i] + 1e-7;
        for (int j = 0; j < 2; ++j) g_rot[i * 2 + j] = (float)((double)g_rot[i * 2 + j] / d);
        for (int j = 0; j < 3; ++j) g_scale[i * 3 + j] = (float)((double)g_scale[i * 3 + j] / d);
    }
}

/* hv_cuda_kernel.cu:183-260.  Outputs must be zero-filled by the caller
 * (zeros_like at :278-280).  Reproduces the reference as-is, including the
 * missing 1/res chain-rule factor on dgrid_dcenter (:219-243 vs :198). */
void hv_oracle_backward(const float* grad, const float* points, const float* xyz,
                        const float* scale, const float* obj, int64_t n, float res,
                        int num_rots, const float corner[3], const int dims[3],
                        float* d_xyz, float* d_scale, float* d_obj) {
    const int X = dims[0], Y = dims[1], Z = dims[2];
    float cs[2 * 4096];
    if (num_rots > 4096) return;
    hv_oracle_rot_table(num_rots, cs);
    for (int64_t c = 0; c < n; ++c) {
        const float objness = obj[c];
        const float cx = xyz[c * 3 + 0] * scale[c * 3 + 0];
        const float cy = xyz[c * 3 + 1] * scale[c * 3 + 1];
        const float cz = xyz[c * 3 + 2] * scale[c * 3 + 2];
        const float px = points[c * 3 + 0], py = points[c * 3 + 1], pz = points[c * 3 + 2];
        for (int i = 0; i < num_rots; ++i) {
            const float ct = cs[2 * i], st = cs[2 * i + 1];
            const float ox = (-ct) * cx + st * cz;
            const float oy = -cy;
            const float oz = (-st) * cx - ct * cz;
            const float gx = ((px + ox) - corner[0]) / res;
            const float gy = ((py + oy) - corner[1]) / res;
            const float gz = ((pz + oz) - corner[2]) / res;
            if (gx < 0 || gy < 0 || gz < 0 || gx >= (float)(X - 1) || gy >= (float)(Y - 1) ||
                gz >= (float)(Z - 1))
                continue;
            const int fx = (int)gx, fy = (int)gy, fz = (int)gz;
            const int hx = fx + 1, hy = fy + 1, hz = fz + 1;
            const float rx = gx - floorf(gx), ry = gy - floorf(gy), rz = gz - floorf(gz);
            const float w0x = 1.f - rx, w0y = 1.f - ry, w0z = 1.f - rz;
            const float w1x = rx, w1y = ry, w1z = rz;
            const float lll = grad[IDX3(fx, fy, fz, Y, Z)], llh = grad[IDX3(fx, fy, hz, Y, Z)];
            const float lhl = grad[IDX3(fx, hy, fz, Y, Z)], lhh = grad[IDX3(fx, hy, hz, Y, Z)];
            const float hll = grad[IDX3(hx, fy, fz, Y, Z)], hlh = grad[IDX3(hx, fy, hz, Y, Z)];
            const float hhl = grad[IDX3(hx, hy, fz, Y, Z)], hhh = grad[IDX3(hx, hy, hz, Y, Z)];
            /* :210-217, eight sequential += into d_obj[c] */
            float dob = d_obj[c];
            dob += lll * w0x * w0y * w0z;
            dob += llh * w0x * w0y * w1z;
            dob += lhl * w0x * w1y * w0z;
            dob += lhh * w0x * w1y * w1z;
            dob += hll * w1x * w0y * w0z;
            dob += hlh * w1x * w0y * w1z;
            dob += hhl * w1x * w1y * w0z;
            dob += hhh * w1x * w1y * w1z;
            d_obj[c] = dob;
            /* :219-243, left-to-right */
            float dx = -lll * w0y * w0z;
            dx = dx - llh * w0y * w1z;
            dx = dx - lhl * w1y * w0z;
            dx = dx - lhh * w1y * w1z;
            dx = dx + hll * w0y * w0z;
            dx = dx + hlh * w0y * w1z;
            dx = dx + hhl * w1y * w0z;
            dx = dx + hhh * w1y * w1z;
            float dy = -lll * w0x * w0z;
            dy = dy - llh * w0x * w1z;
            dy = dy + lhl * w0x * w0z;
            dy = dy + lhh * w0x * w1z;
            dy = dy - hll * w1x * w0z;
            dy = dy - hlh * w1x * w1z;
            dy = dy + hhl * w1x * w0z;
            dy = dy + hhh * w1x * w1z;
            float dz = -lll * w0x * w0y;
            dz = dz + llh * w0x * w0y;
            dz = dz - lhl * w0x * w1y;
            dz = dz + lhh * w0x * w1y;
            dz = dz - hll * w1x * w0y;
            dz = dz + hlh * w1x * w0y;
            dz = dz - hhl * w1x * w1y;
            dz = dz + hhh * w1x * w1y;
            dx = dx * objness; dy = dy * objness; dz = dz * objness;
            /* :249-250 */
            const float dcx = (-ct) * dx - st * dz;
            const float dcy = -dy;
            const float dcz = st * dx - ct * dz;
            /* :252-258 */
            d_xyz[c * 3 + 0] += dcx * scale[c * 3 + 0];
            d_xyz[c * 3 + 1] += dcy * scale[c * 3 + 1];
            d_xyz[c * 3 + 2] += dcz * scale[c * 3 + 2];
            d_scale[c * 3 + 0] += dcx * xyz[c * 3 + 0];
            d_scale[c * 3 + 1] += dcy * xyz[c * 3 + 1];
            d_scale[c * 3 + 2] += dcz * xyz[c * 3 + 2];
        }
    }
}

/* ---- what-if variants of the vote geometry (tests/test_vote_contraction.py) -----------------------------------
 * The CUDA binary of the reference cannot be produced here, and two of its choices are invisible in the source:
 * nvcc contracts `-cos*cx + sin*cz` and `-sin*cx - cos*cz` (hv_cuda_kernel.cu:38-39) into FMAs (-fmad=true is its
 * default) and calls its own cosf/sinf (<= 2 ulp).  The functions below evaluate the SAME statements under those
 * alternatives so that a CPU test can count how many votes would land in a different cell / change their in-bounds
 * status, and whether any decode output changes.
 *   fma_mode 0: no contraction (the oracle's and the HIP kernel's convention)
 *            1: the SECOND product of each sum is the fused one: fma(s, cz, (-c)*cx), fma(-c, cz, (-s)*cx)
 *            2: the FIRST product is the fused one (LLVM's (fadd (fmul x, y), z) -> fma x, y, z rule, the likely nvcc
 *               output): fma(-c, cx, s*cz), fma(-s, cx, -(c*cz))
 *   cs: [num_rots][2] table of (cos, sin) to use (any <= 2 ulp cosf/sinf), or NULL for the oracle's table. */
static inline void hv_variant_offset(float ct, float st, float cx, float cz, int fma_mode, float* ox, float* oz) {
    if (fma_mode == 1) {
        *ox = fmaf(st, cz, (-ct) * cx);
        *oz = fmaf(-ct, cz, (-st) * cx);
    } else if (fma_mode == 2) {
        *ox = fmaf(-ct, cx, st * cz);
        *oz = fmaf(-st, cx, -(ct * cz));
    } else {
        *ox = (-ct) * cx + st * cz;
        *oz = (-st) * cx - ct * cz;
    }
}

/* out[0] votes examined, out[1] in-bounds under A, out[2] in-bounds status differs, out[3] both in bounds but another
 * floor cell, out[4] points with at least one changed vote */
void hv_oracle_vote_diff(const float* points, const float* xyz, const float* scale, int64_t n, float res, int num_rots,
                         const float corner[3], const int dims[3], const float* cs_a, int mode_a, const float* cs_b,
                         int mode_b, int64_t out[5]) {
    float cs0[2 * 4096];
    const int X = dims[0], Y = dims[1], Z = dims[2];
    for (int k = 0; k < 5; ++k) out[k] = 0;
    if (num_rots > 4096) return;
    hv_oracle_rot_table(num_rots, cs0);
    if (!cs_a) cs_a = cs0;
    if (!cs_b) cs_b = cs0;
    for (int64_t c = 0; c < n; ++c) {
        const float cx = xyz[c * 3 + 0] * scale[c * 3 + 0];
        const float cy = xyz[c * 3 + 1] * scale[c * 3 + 1];
        const float cz = xyz[c * 3 + 2] * scale[c * 3 + 2];
        const float px = points[c * 3 + 0], py = points[c * 3 + 1], pz = points[c * 3 + 2];
        int changed = 0;
        for (int i = 0; i < num_rots; ++i) {
            float g[2][3];
            int in[2];
            for (int v = 0; v < 2; ++v) {
                const float* cs = v ? cs_b : cs_a;
                float ox, oz;
                hv_variant_offset(cs[2 * i], cs[2 * i + 1], cx, cz, v ? mode_b : mode_a, &ox, &oz);
                g[v][0] = ((px + ox) - corner[0]) / res;
                g[v][1] = ((py + (-cy)) - corner[1]) / res;
                g[v][2] = ((pz + oz) - corner[2]) / res;
                in[v] = !(g[v][0] < 0 || g[v][1] < 0 || g[v][2] < 0 || g[v][0] >= (float)(X - 1) ||
                          g[v][1] >= (float)(Y - 1) || g[v][2] >= (float)(Z - 1));
            }
            ++out[0];
            out[1] += in[0];
            if (in[0] != in[1]) { ++out[2]; changed = 1; }
            else if (in[0] && ((int)g[0][0] != (int)g[1][0] || (int)g[0][1] != (int)g[1][1] || (int)g[0][2] != (int)g[1][2])) {
                ++out[3];
                changed = 1;
            }
        }
        out[4] += changed;
    }
}

/* hv_oracle_forward with the geometry variant (grids zero-filled by the caller; everything after the offset is the
 * oracle's own sequence) */
int64_t hv_oracle_forward_variant(const float* points, const float* xyz, const float* scale, const float* obj, int64_t n,
                                  float res, int num_rots, const float corner[3], const int dims[3], const float* cs_in,
                                  int fma_mode, float* g_obj, float* g_rot, float* g_scale) {
    const int X = dims[0], Y = dims[1], Z = dims[2];
    float cs0[2 * 4096];
    if (num_rots > 4096) return -1;
    hv_oracle_rot_table(num_rots, cs0);
    const float* cs = cs_in ? cs_in : cs0;
    int64_t v_in = 0;
    for (int64_t c = 0; c < n; ++c) {
        const float objness = obj[c];
        const float cx = xyz[c * 3 + 0] * scale[c * 3 + 0];
        const float cy = xyz[c * 3 + 1] * scale[c * 3 + 1];
        const float cz = xyz[c * 3 + 2] * scale[c * 3 + 2];
        const float px = points[c * 3 + 0], py = points[c * 3 + 1], pz = points[c * 3 + 2];
        for (int i = 0; i < num_rots; ++i) {
            const float ct = cs[2 * i], st = cs[2 * i + 1];
            float ox, oz;
            hv_variant_offset(ct, st, cx, cz, fma_mode, &ox, &oz);
            const float oy = -cy;
            const float gx = ((px + ox) - corner[0]) / res;
            const float gy = ((py + oy) - corner[1]) / res;
            const float gz = ((pz + oz) - corner[2]) / res;
            if (gx < 0 || gy < 0 || gz < 0 || gx >= (float)(X - 1) || gy >= (float)(Y - 1) || gz >= (float)(Z - 1)) continue;
            ++v_in;
            const int fx = (int)gx, fy = (int)gy, fz = (int)gz;
            const int hx = fx + 1, hy = fy + 1, hz = fz + 1;
            const float rx = gx - floorf(gx), ry = gy - floorf(gy), rz = gz - floorf(gz);
            const float w0x = 1.f - rx, w0y = 1.f - ry, w0z = 1.f - rz;
            const float w[8] = {w0x * w0y * w0z * objness, w0x * w0y * rz * objness, w0x * ry * w0z * objness,
                                w0x * ry * rz * objness,   rx * w0y * w0z * objness, rx * w0y * rz * objness,
                                rx * ry * w0z * objness,   rx * ry * rz * objness};
            const int64_t cell[8] = {IDX3(fx, fy, fz, Y, Z), IDX3(fx, fy, hz, Y, Z), IDX3(fx, hy, fz, Y, Z),
                                     IDX3(fx, hy, hz, Y, Z), IDX3(hx, fy, fz, Y, Z), IDX3(hx, fy, hz, Y, Z),
                                     IDX3(hx, hy, fz, Y, Z), IDX3(hx, hy, hz, Y, Z)};
            for (int k = 0; k < 8; ++k) g_obj[cell[k]] += w[k];
            const float rot_vec[2] = {ct, st};
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 8; ++k) g_rot[cell[k] * 2 + j] += w[k] * rot_vec[j];
            for (int j = 0; j < 3; ++j) {
                const float s = scale[c * 3 + j];
                for (int k = 0; k < 8; ++k) g_scale[cell[k] * 3 + j] += w[k] * s;
            }
        }
    }
    return v_in;
}
