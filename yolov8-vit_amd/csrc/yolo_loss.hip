// v8 detection loss + gradient on the device (the loss that
// `YOLO(pt).train(...)` optimises, utils/trainYolo.py:33; all of it lives inside `ultralytics`, absent from the tree:
// PARITY UNPINNED, restated from the published v8DetectionLoss in oracle/yolo_train.py):
//   DFL decode (reg_max 16) -> TaskAlignedAssigner(topk 10, alpha 0.5, beta 6) -> CIoU box loss + DFL + BCE class loss,
//   gains 7.5 / 0.5 / 1.5, normalised by the batch sum of target scores, times the batch size.
// Build-defined where the published code leaves it to torch.topk's tie order: the top-k of a ground truth are taken
// among the anchors whose centre lies inside its box, ties (e.g. metric 0) by ascending anchor index.
// Latency-bound bookkeeping (A = 8400 anchors x G boxes per image): thread per anchor, workgroup per (image, box) for the
// selections; every reduction is two-stage in a fixed order (bitwise reproducible).  Counters use integer atomics only.
#include "yv_common.h"

namespace {

constexpr int RM = 16;            // reg_max
constexpr int TOPK = 10;
constexpr int MAX_NC = 80;

struct LossArgs {
    const float* box[3];          // (B*h*h, 64) logits per scale
    const float* cls[3];          // (B*h*h, ncp)
    float* dbox[3];
    float* dcls[3];
    int hw[3];                    // h (= w) per scale
    int off[4];                   // anchor offsets of the scales, off[3] = A
    int stride[3];
    int B, A, G, nc, ncp;
    const float* gt_box;          // (B, G, 4) xyxy input pixels
    const int32_t* gt_lab;        // (B, G)
    const int32_t* gt_n;          // (B)
    // scratch
    float* pb;                    // (B, A, 4) predicted xyxy in grid units
    float* iou;                   // (B, G, A)
    float* align;                 // (B, G, A)
    int32_t* cnt;                 // (B, A)
    int32_t* gsel;                // (B, A)
    int32_t* tgt;                 // (B, A)
    float* norm;                  // (B, A)
    float* pos;                   // (B, G, 2): max align, max iou over the positives of a box
    float* part;                  // partial sums
    float* tss;                   // (1)
    float* loss;                  // (4): total*B, box, cls, dfl
    float gain_box, gain_cls, gain_dfl;
};

__device__ __forceinline__ void anchor_of(const LossArgs& a, int n, int& s, int& row_in_img, float& ax, float& ay) {
    s = n >= a.off[2] ? 2 : (n >= a.off[1] ? 1 : 0);
    row_in_img = n - a.off[s];
    const int w = a.hw[s];
    const int y = row_in_img / w, x = row_in_img - y * w;
    ax = x + 0.5f; ay = y + 0.5f;
}

struct Ciou { float v, iou; };

// CIoU of prediction p (x1,y1,x2,y2) against target t; optional gradient wrt p (alpha constant, as published)
__device__ __forceinline__ float ciou_f(const float* p, const float* t, float* grad) {
    const float eps = 1e-7f;
    const float w1 = p[2] - p[0], h1 = p[3] - p[1] + eps, w2 = t[2] - t[0], h2 = t[3] - t[1] + eps;
    const float ix1 = fmaxf(p[0], t[0]), ix2 = fminf(p[2], t[2]), iy1 = fmaxf(p[1], t[1]), iy2 = fminf(p[3], t[3]);
    const float iw = fmaxf(ix2 - ix1, 0.f), ih = fmaxf(iy2 - iy1, 0.f);
    const float inter = iw * ih;
    const float uni = w1 * h1 + w2 * h2 - inter + eps;
    const float iou = inter / uni;
    const float cw = fmaxf(p[2], t[2]) - fminf(p[0], t[0]), ch = fmaxf(p[3], t[3]) - fminf(p[1], t[1]);
    const float c2 = cw * cw + ch * ch + eps;
    const float sx = t[0] + t[2] - p[0] - p[2], sy = t[1] + t[3] - p[1] - p[3];
    const float rho2 = (sx * sx + sy * sy) * 0.25f;
    const float da = atanf(w2 / h2) - atanf(w1 / h1);
    const float k4 = 0.40528473456935109f;                        // 4 / pi^2
    const float v = k4 * da * da;
    const float alpha = v / (v - iou + (1.0f + eps));
    const float ciou = iou - (rho2 / c2 + v * alpha);
    if (grad) {
        // d inter, d area1, d cw/ch, d rho2, d atan(w1/h1) per coordinate (x1, y1, x2, y2)
        const float diw[4] = {(iw > 0.f && p[0] > t[0]) ? -1.f : 0.f, 0.f, (iw > 0.f && p[2] < t[2]) ? 1.f : 0.f, 0.f};
        const float dih[4] = {0.f, (ih > 0.f && p[1] > t[1]) ? -1.f : 0.f, 0.f, (ih > 0.f && p[3] < t[3]) ? 1.f : 0.f};
        const float dar[4] = {-h1, -w1, h1, w1};
        const float dcw[4] = {p[0] < t[0] ? -1.f : 0.f, 0.f, p[2] > t[2] ? 1.f : 0.f, 0.f};
        const float dch[4] = {0.f, p[1] < t[1] ? -1.f : 0.f, 0.f, p[3] > t[3] ? 1.f : 0.f};
        const float drho[4] = {-0.5f * sx, -0.5f * sy, -0.5f * sx, -0.5f * sy};
        const float den = w1 * w1 + h1 * h1;
        const float dat[4] = {-h1 / den, w1 / den, h1 / den, -w1 / den};       // d atan(w1/h1): dw1 = -+1, dh1 = -+1
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dinter = diw[i] * ih + iw * dih[i];
            const float duni = dar[i] - dinter;
            const float diou = (dinter * uni - inter * duni) / (uni * uni);
            const float dc2 = 2.f * cw * dcw[i] + 2.f * ch * dch[i];
            const float dpen = (drho[i] * c2 - rho2 * dc2) / (c2 * c2);
            const float dv = 2.f * k4 * da * (-dat[i]);
            grad[i] = diou - dpen - alpha * dv;
        }
    }
    return ciou;
}

// ---- A: decode + per (box, anchor) overlap / alignment metric ------------------------------------------------
__global__ __launch_bounds__(256) void loss_metrics_kernel(LossArgs a) {
    const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (n >= a.A) return;
    int s, r; float ax, ay;
    anchor_of(a, n, s, r, ax, ay);
    const long long row = (long long)b * a.hw[s] * a.hw[s] + r;
    const float* bl = a.box[s] + row * 64;
    float d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float m = bl[k * RM];
        for (int i = 1; i < RM; ++i) m = fmaxf(m, bl[k * RM + i]);
        float den = 0.f, num = 0.f;
        for (int i = 0; i < RM; ++i) { const float e = __expf(bl[k * RM + i] - m); den += e; num += e * (float)i; }
        d[k] = num / den;
    }
    float pb[4] = {ax - d[0], ay - d[1], ax + d[2], ay + d[3]};
    float* pbo = a.pb + ((long long)b * a.A + n) * 4;
    pbo[0] = pb[0]; pbo[1] = pb[1]; pbo[2] = pb[2]; pbo[3] = pb[3];
    const float st = (float)a.stride[s];
    const float px = ax * st, py = ay * st;
    const float pbp[4] = {pb[0] * st, pb[1] * st, pb[2] * st, pb[3] * st};
    if (n < a.A) { a.cnt[(long long)b * a.A + n] = 0; a.gsel[(long long)b * a.A + n] = -1; }
    const int ng = a.gt_n[b];
    const float* cl = a.cls[s] + row * a.ncp;
    for (int g = 0; g < ng; ++g) {
        const float* gb = a.gt_box + ((long long)b * a.G + g) * 4;
        const float dm = fminf(fminf(px - gb[0], py - gb[1]), fminf(gb[2] - px, gb[3] - py));
        float iou = 0.f, al = 0.f;
        if (dm > 1e-9f) {
            const float t[4] = {gb[0], gb[1], gb[2], gb[3]};
            iou = fmaxf(ciou_f(t, pbp, nullptr), 0.f);
            int lab = a.gt_lab[(long long)b * a.G + g];
            lab = lab < 0 ? 0 : (lab >= a.nc ? a.nc - 1 : lab);
            const float sc = 1.0f / (1.0f + __expf(-cl[lab]));
            const float i2 = iou * iou;
            al = sqrtf(sc) * (i2 * i2 * i2);
        }
        const long long o = ((long long)b * a.G + g) * a.A + n;
        a.iou[o] = iou; a.align[o] = al;
    }
}

// ---- B: top-k anchors of every ground truth (inside its box; ties by ascending anchor index) ----------------------
__global__ __launch_bounds__(256) void loss_topk_kernel(LossArgs a) {
    const int g = blockIdx.x, b = blockIdx.y;
    if (g >= a.gt_n[b]) return;
    __shared__ float sv[256];
    __shared__ int si[256];
    __shared__ int chosen[TOPK];
    const float* gb = a.gt_box + ((long long)b * a.G + g) * 4;
    const float* al = a.align + ((long long)b * a.G + g) * a.A;
    for (int k = 0; k < TOPK; ++k) {
        float best = -1.f; int bi = 0x7fffffff;
        for (int n = threadIdx.x; n < a.A; n += 256) {
            int s, r; float ax, ay;
            anchor_of(a, n, s, r, ax, ay);
            const float st = (float)a.stride[s];
            const float px = ax * st, py = ay * st;
            const float dm = fminf(fminf(px - gb[0], py - gb[1]), fminf(gb[2] - px, gb[3] - py));
            if (!(dm > 1e-9f)) continue;
            bool taken = false;
            for (int q = 0; q < k; ++q) taken |= chosen[q] == n;
            if (taken) continue;
            const float v = al[n];
            if (v > best || (v == best && n < bi)) { best = v; bi = n; }
        }
        sv[threadIdx.x] = best; si[threadIdx.x] = bi;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) {
                const float v2 = sv[threadIdx.x + o]; const int i2 = si[threadIdx.x + o];
                if (v2 > sv[threadIdx.x] || (v2 == sv[threadIdx.x] && i2 < si[threadIdx.x])) { sv[threadIdx.x] = v2; si[threadIdx.x] = i2; }
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            chosen[k] = sv[0] >= 0.f ? si[0] : -1;
            if (sv[0] >= 0.f) {
                atomicAdd(&a.cnt[(long long)b * a.A + si[0]], 1);
                a.gsel[(long long)b * a.A + si[0]] = g;
            }
        }
        __syncthreads();
        if (chosen[k] < 0) break;                     // fewer than k anchors inside the box
    }
}

// ---- C: one ground truth per anchor (several claimants: highest overlap, first on ties) -----------------------------
__global__ __launch_bounds__(256) void loss_resolve_kernel(LossArgs a) {
    const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (n >= a.A) return;
    const long long o = (long long)b * a.A + n;
    const int c = a.cnt[o];
    int t = -1;
    if (c == 1) t = a.gsel[o];
    else if (c > 1) {
        float best = -1.f;
        const int ng = a.gt_n[b];
        for (int g = 0; g < ng; ++g) {
            const float v = a.iou[((long long)b * a.G + g) * a.A + n];
            if (v > best) { best = v; t = g; }
        }
    }
    a.tgt[o] = t;
}

// ---- D: per ground truth, maxima of metric / overlap over its positives ---------------------------------------
__global__ __launch_bounds__(256) void loss_posmax_kernel(LossArgs a) {
    const int g = blockIdx.x, b = blockIdx.y;
    if (g >= a.gt_n[b]) return;
    __shared__ float s0[256], s1[256];
    float m0 = 0.f, m1 = 0.f;
    for (int n = threadIdx.x; n < a.A; n += 256) {
        if (a.tgt[(long long)b * a.A + n] != g) continue;
        const long long o = ((long long)b * a.G + g) * a.A + n;
        m0 = fmaxf(m0, a.align[o]); m1 = fmaxf(m1, a.iou[o]);
    }
    s0[threadIdx.x] = m0; s1[threadIdx.x] = m1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { s0[threadIdx.x] = fmaxf(s0[threadIdx.x], s0[threadIdx.x + o]); s1[threadIdx.x] = fmaxf(s1[threadIdx.x], s1[threadIdx.x + o]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { a.pos[((long long)b * a.G + g) * 2] = s0[0]; a.pos[((long long)b * a.G + g) * 2 + 1] = s1[0]; }
}

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---- E: normalised target score of every anchor + partial sums ----------------------------------------------------
__global__ __launch_bounds__(256) void loss_norm_kernel(LossArgs a) {
    __shared__ float sh[4];
    const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    float v = 0.f;
    if (n < a.A) {
        const long long o = (long long)b * a.A + n;
        const int t = a.tgt[o];
        if (t >= 0) {
            const float* ps = a.pos + ((long long)b * a.G + t) * 2;
            v = a.align[((long long)b * a.G + t) * a.A + n] * ps[1] / (ps[0] + 1e-9f);
        }
        a.norm[o] = v;
    }
    const float s = block_sum_256(v, sh);
    if (threadIdx.x == 0) a.part[(long long)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

__global__ __launch_bounds__(64) void loss_tss_kernel(LossArgs a, int nparts) {
    if (threadIdx.x) return;
    float s = 0.f;
    for (int i = 0; i < nparts; ++i) s += a.part[i];
    a.tss[0] = fmaxf(s, 1.0f);
}

// ---- F: losses and gradients -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_grad_kernel(LossArgs a) {
    __shared__ float sh[4];
    const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    float l_box = 0.f, l_cls = 0.f, l_dfl = 0.f;
    if (n < a.A) {
        int s, r; float ax, ay;
        anchor_of(a, n, s, r, ax, ay);
        const long long row = (long long)b * a.hw[s] * a.hw[s] + r;
        const long long o = (long long)b * a.A + n;
        const float inv = 1.0f / a.tss[0];
        const float Bf = (float)a.B;
        const int t = a.tgt[o];
        const float w = a.norm[o];
        int lab = -1;
        if (t >= 0) { lab = a.gt_lab[(long long)b * a.G + t]; lab = lab < 0 ? 0 : (lab >= a.nc ? a.nc - 1 : lab); }
        const float* cl = a.cls[s] + row * a.ncp;
        float* dc = a.dcls[s] + row * a.ncp;
        for (int c = 0; c < a.ncp; ++c) {
            float gq = 0.f;
            if (c < a.nc) {
                const float x = cl[c], tg = c == lab ? w : 0.f;
                l_cls += fmaxf(x, 0.f) - x * tg + log1pf(__expf(-fabsf(x)));
                gq = (1.0f / (1.0f + __expf(-x)) - tg) * inv * a.gain_cls * Bf;
            }
            dc[c] = gq;
        }
        const float* bl = a.box[s] + row * 64;
        float* db = a.dbox[s] + row * 64;
        if (t < 0) {
            for (int i = 0; i < 64; ++i) db[i] = 0.f;
        } else {
            const float st = (float)a.stride[s];
            const float* gb = a.gt_box + ((long long)b * a.G + t) * 4;
            const float tb[4] = {gb[0] / st, gb[1] / st, gb[2] / st, gb[3] / st};
            const float* pbp = a.pb + o * 4;
            const float pb[4] = {pbp[0], pbp[1], pbp[2], pbp[3]};
            float gci[4];
            const float ci = ciou_f(pb, tb, gci);
            l_box = (1.0f - ci) * w;
            const float gs = -w * inv * a.gain_box * Bf;                       // d L / d ciou
            const float gdist[4] = {-gs * gci[0], -gs * gci[1], gs * gci[2], gs * gci[3]};      // l, t, r, b
            const float tl4[4] = {ax - tb[0], ay - tb[1], tb[2] - ax, tb[3] - ay};
            for (int k = 0; k < 4; ++k) {
                float m = bl[k * RM];
                for (int i = 1; i < RM; ++i) m = fmaxf(m, bl[k * RM + i]);
                float p[RM], den = 0.f, dist = 0.f;
                for (int i = 0; i < RM; ++i) { p[i] = __expf(bl[k * RM + i] - m); den += p[i]; }
                const float lse = m + __logf(den);
                for (int i = 0; i < RM; ++i) { p[i] /= den; dist += p[i] * (float)i; }
                float tt = fminf(fmaxf(tl4[k], 0.f), (float)(RM - 1) - 0.01f);
                const int tl = (int)tt, tr = tl + 1;
                const float wl = (float)tr - tt, wr = 1.0f - wl;
                l_dfl += ((lse - bl[k * RM + tl]) * wl + (lse - bl[k * RM + tr]) * wr) * 0.25f * w;
                const float gd = w * inv * a.gain_dfl * Bf * 0.25f;
                for (int i = 0; i < RM; ++i) {
                    float gq = gdist[k] * p[i] * ((float)i - dist);
                    gq += gd * (p[i] - (i == tl ? wl : 0.f) - (i == tr ? wr : 0.f));
                    db[k * RM + i] = gq;
                }
            }
        }
    }
    const float s0 = block_sum_256(l_box, sh), s1 = block_sum_256(l_cls, sh), s2 = block_sum_256(l_dfl, sh);
    if (threadIdx.x == 0) {
        float* p = a.part + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 3;
        p[0] = s0; p[1] = s1; p[2] = s2;
    }
}

__global__ __launch_bounds__(64) void loss_final_kernel(LossArgs a, int nparts) {
    if (threadIdx.x) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < nparts; ++i) { s0 += a.part[i * 3]; s1 += a.part[i * 3 + 1]; s2 += a.part[i * 3 + 2]; }
    const float inv = 1.0f / a.tss[0];
    s0 *= inv; s1 *= inv; s2 *= inv;
    a.loss[0] = (a.gain_box * s0 + a.gain_cls * s1 + a.gain_dfl * s2) * (float)a.B;
    a.loss[1] = s0; a.loss[2] = s1; a.loss[3] = s2;
}

}  // namespace

extern "C" size_t yv_detect_loss_ws_bytes(int B, int A, int G) {
    const size_t blocks = (size_t)((A + 255) / 256) * B;
    size_t f = (size_t)B * A * 4 + 2 * (size_t)B * G * A + (size_t)B * A + (size_t)B * G * 2 + blocks * 3 + 8;   // floats
    size_t i = 3 * (size_t)B * A;                                                                                   // ints
    return (f + i) * 4 + 256;
}

extern "C" int yv_detect_loss(const float* const* box, const float* const* cls, float* const* dbox, float* const* dcls,
                              int B, int size, int nc, int ncp, const float* gt_boxes, const int32_t* gt_labels,
                              const int32_t* gt_counts, int G, float gain_box, float gain_cls, float gain_dfl, float* loss,
                              void* ws, size_t ws_bytes, void* stream) {
    if (!box || !cls || !dbox || !dcls || !gt_boxes || !gt_labels || !gt_counts || !loss || !ws) return YV_ERR_ARG;
    if (B <= 0 || size <= 0 || (size % 32) || nc <= 0 || nc > MAX_NC || ncp < nc || G <= 0) return YV_ERR_ARG;
    LossArgs a = {};
    int A = 0;
    for (int s = 0; s < 3; ++s) {
        if (!box[s] || !cls[s] || !dbox[s] || !dcls[s]) return YV_ERR_ARG;
        a.box[s] = box[s]; a.cls[s] = cls[s]; a.dbox[s] = dbox[s]; a.dcls[s] = dcls[s];
        a.stride[s] = 8 << s; a.hw[s] = size / a.stride[s]; a.off[s] = A; A += a.hw[s] * a.hw[s];
    }
    a.off[3] = A;
    if (ws_bytes < yv_detect_loss_ws_bytes(B, A, G)) return YV_ERR_WORKSPACE;
    a.B = B; a.A = A; a.G = G; a.nc = nc; a.ncp = ncp;
    a.gt_box = gt_boxes; a.gt_lab = gt_labels; a.gt_n = gt_counts;
    a.gain_box = gain_box; a.gain_cls = gain_cls; a.gain_dfl = gain_dfl; a.loss = loss;
    const unsigned bx = (unsigned)((A + 255) / 256);
    float* f = (float*)ws;
    a.pb = f; f += (size_t)B * A * 4;
    a.iou = f; f += (size_t)B * G * A;
    a.align = f; f += (size_t)B * G * A;
    a.norm = f; f += (size_t)B * A;
    a.pos = f; f += (size_t)B * G * 2;
    a.part = f; f += (size_t)bx * B * 3;
    a.tss = f; f += 8;
    int32_t* ip = (int32_t*)f;
    a.cnt = ip; ip += (size_t)B * A;
    a.gsel = ip; ip += (size_t)B * A;
    a.tgt = ip;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_metrics_kernel, dim3(bx, B), dim3(256), 0, st, a);
    hipLaunchKernelGGL(loss_topk_kernel, dim3(G, B), dim3(256), 0, st, a);
    hipLaunchKernelGGL(loss_resolve_kernel, dim3(bx, B), dim3(256), 0, st, a);
    hipLaunchKernelGGL(loss_posmax_kernel, dim3(G, B), dim3(256), 0, st, a);
    hipLaunchKernelGGL(loss_norm_kernel, dim3(bx, B), dim3(256), 0, st, a);
    hipLaunchKernelGGL(loss_tss_kernel, dim3(1), dim3(64), 0, st, a, (int)(bx * B));
    hipLaunchKernelGGL(loss_grad_kernel, dim3(bx, B), dim3(256), 0, st, a);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, a, (int)(bx * B));
    return yv_launch_status();
}
