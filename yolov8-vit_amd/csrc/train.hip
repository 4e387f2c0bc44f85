// Fine-tune scalar-side kernels.
//   yv_loss_fwd_bwd  build_loss = LSCE(0.1)/6 + Focal(alpha 1, gamma 2, mean)*5/6
//                    utils/trainClass.py:46-66,162-185,362-370  (forward value + d/dlogits)
//   yv_sgd_step      torch.optim.SGD(lr, momentum=0.9, weight_decay=1e-3)   utils/trainClass.py:442-443
//                    (grad_scale = 1/world folds the data-parallel gradient mean into the update)
// The loss is (B,5): latency-bound, one workgroup.  SGD is a 20 B/param HBM stream
// (read p,g,m; write p,m) issued as 16-byte accesses.
#include "yv_common.h"

namespace {

constexpr int LOSS_MAX_NC = 32;

__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                   int B, int nc, float w_ls, float w_fo, float* __restrict__ loss,
                                                   float* __restrict__ grad) {
    __shared__ float red[4];
    float acc = 0.f;
    const float inv_b = 1.0f / (float)B, inv_bc = 1.0f / ((float)B * (float)nc), inv_c = 1.0f / (float)nc;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const float* x = logits + (size_t)i * nc;
        const int y = labels[i];
        float m = x[0];
        for (int c = 1; c < nc; ++c) m = fmaxf(m, x[c]);
        float e[LOSS_MAX_NC], den = 0.f;
        for (int c = 0; c < nc; ++c) { e[c] = expf(x[c] - m); den += e[c]; }
        float cross = 0.f, smooth = 0.f, fl = 0.f;
        for (int c = 0; c < nc; ++c) {
            const float p = e[c] / den;                       // y_hat = softmax(x)
            const float nl = -logf(p);                        // -log(y_hat)
            smooth += nl;
            if (c == y) cross = nl;
            const float t = c == y ? 1.f : 0.f;
            const float xc = x[c];
            const float bce = fmaxf(xc, 0.f) - xc * t + log1pf(expf(-fabsf(xc)));
            const float pt = expf(-bce);
            const float om = 1.f - pt;
            fl += om * om * bce;
            const float sig = 1.f / (1.f + expf(-xc));
            const float g_ls = (p - 0.9f * t - 0.1f * inv_c) * inv_b;
            const float g_fo = (sig - t) * om * (2.f * pt * bce + om) * inv_bc;
            grad[(size_t)i * nc + c] = g_ls * w_ls + g_fo * w_fo;
        }
        acc += (0.9f * cross + 0.1f * smooth * inv_c) * inv_b * w_ls + fl * inv_bc * w_fo;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ m, size_t n, float lr, float mu, float wd,
                                                  float gs, int first, uint16_t* __restrict__ mirror) {
    const size_t n4 = n >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pv = ((float4*)p)[i], gv = ((const float4*)g)[i], mv;
        float pe[4] = {pv.x, pv.y, pv.z, pv.w}, ge[4] = {gv.x, gv.y, gv.z, gv.w}, me[4];
        if (!first) { mv = ((float4*)m)[i]; me[0] = mv.x; me[1] = mv.y; me[2] = mv.z; me[3] = mv.w; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float gq = __fadd_rn(__fmul_rn(ge[q], gs), __fmul_rn(wd, pe[q]));
            me[q] = first ? gq : __fadd_rn(__fmul_rn(mu, me[q]), gq);
            pe[q] = __fsub_rn(pe[q], __fmul_rn(lr, me[q]));
        }
        ((float4*)p)[i] = make_float4(pe[0], pe[1], pe[2], pe[3]);
        ((float4*)m)[i] = make_float4(me[0], me[1], me[2], me[3]);
        if (mirror) ((uint2*)mirror)[i] = make_uint2(pack_bf16x2(pe[0], pe[1]), pack_bf16x2(pe[2], pe[3]));   // bf16 working copy
    }
    // tail (< 4 elements)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        size_t i = (n4 << 2) + threadIdx.x;
        float gq = __fadd_rn(__fmul_rn(g[i], gs), __fmul_rn(wd, p[i]));
        float mq = first ? gq : __fadd_rn(__fmul_rn(mu, m[i]), gq);
        m[i] = mq;
        p[i] = __fsub_rn(p[i], __fmul_rn(lr, mq));
        if (mirror) mirror[i] = f32_to_bf16(p[i]);
    }
}

// Optimisers of the detector trainer (what ultralytics builds for `optimizer='auto'`: torch.optim.AdamW for short runs,
// torch.optim.SGD(nesterov=True) for long ones) + the ModelEMA update.  One element per thread, explicit _rn ops in the
// operation order of torch's single-tensor implementations.
//   kind 1 SGD-Nesterov: g += wd*p; buf = first ? g : mu*buf + g; p -= lr*(g + mu*buf)
//   kind 2 AdamW:        p *= 1 - lr*wd; m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g;
//                        p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void optim_kernel(int kind, float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, size_t n, float lr, float b1,
                                                    float b2, float eps, float wd, float gs, float decay_mul, float om_b1,
                                                    float om_b2, float step_size, float sqrt_bc2, int first,
                                                    uint16_t* __restrict__ mirror) {
    // decay_mul = 1 - lr*wd, om_b* = 1 - beta*, step_size = lr / (1 - beta1^t), sqrt_bc2 = sqrt(1 - beta2^t): evaluated in
    // double on the host and rounded once, as torch's Python scalars are
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float pe = p[i];
        float ge = gs == 1.0f ? g[i] : __fmul_rn(g[i], gs);
        if (kind == 1) {
            ge = __fadd_rn(ge, __fmul_rn(wd, pe));
            const float buf = first ? ge : __fadd_rn(__fmul_rn(b1, m[i]), ge);
            m[i] = buf;
            pe = __fsub_rn(pe, __fmul_rn(lr, __fadd_rn(ge, __fmul_rn(b1, buf))));
        } else {
            pe = __fmul_rn(pe, decay_mul);
            const float mo = first ? 0.f : m[i], vo = first ? 0.f : v[i];
            const float mn = __fadd_rn(mo, __fmul_rn(om_b1, __fsub_rn(ge, mo)));                       // exp_avg.lerp_(grad, 1 - b1)
            const float vn = __fadd_rn(__fmul_rn(vo, b2), __fmul_rn(__fmul_rn(om_b2, ge), ge));       // mul_(b2).addcmul_(g, g, 1 - b2)
            m[i] = mn; v[i] = vn;
            const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vn), sqrt_bc2), eps);
            pe = __fsub_rn(pe, __fmul_rn(step_size, __fdiv_rn(mn, denom)));                             // addcdiv_(m, denom, -step_size)
        }
        p[i] = pe;
        if (mirror) mirror[i] = f32_to_bf16(pe);
    }
}

__global__ __launch_bounds__(256) void axpby_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n, float a,
                                                    float b) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = __fadd_rn(__fmul_rn(dst[i], a), __fmul_rn(b, src[i]));            // ModelEMA: v *= d; v += (1 - d) * model
}

// Batched bf16 transpose dst[b][c][r] = src[b][r][c] (64 x 64 tiles through LDS, 16-byte global accesses): the trainer keeps a
// TRANSPOSED bf16 mirror of every block linear's weight, so that the data gradient dX = dY . W is an ordinary "NT" product
// (out columns = the weight's input features) and runs on the persistent forward GEMM instead of the transposing-read kernel.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int rows,
                                                            int cols, long long src_stride, long long dst_stride) {
    __shared__ uint16_t tile[64][64 + 2];
    const uint16_t* s = src + (long long)blockIdx.z * src_stride;
    uint16_t* d = dst + (long long)blockIdx.z * dst_stride;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                // 64 rows x 8 chunks of 8 elements
        const int idx = t + i * 256, r = idx >> 3, ch = idx & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + r < rows && c0 + ch * 8 < cols) v = *(const uint4*)(s + (long long)(r0 + r) * cols + c0 + ch * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[r][ch * 8 + e] = (uint16_t)(w[e >> 1] >> ((e & 1) * 16));
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = t + i * 256, c = idx >> 3, ch = idx & 7;  // output row c (a source column), 8 source rows per chunk
        if (c0 + c < cols && r0 + ch * 8 < rows) {
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (uint32_t)tile[ch * 8 + 2 * e][c] | ((uint32_t)tile[ch * 8 + 2 * e + 1][c] << 16);
            *(uint4*)(d + (long long)(c0 + c) * rows + r0 + ch * 8) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

}  // namespace

extern "C" int yv_optim_step(int kind, float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, float grad_scale, int step, void* bf16_mirror,
                             void* stream) {
    if (!(kind == 1 || kind == 2) || !p || !g || !m || (kind == 2 && !v) || step < 1) return YV_ERR_ARG;
    if (n == 0) return YV_OK;
    const size_t want = (n + 255) / 256;
    const int blocks = (int)(want > 4096 ? 4096 : want);
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(optim_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, kind, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, grad_scale, (float)(1.0 - (double)lr * (double)weight_decay), (float)(1.0 - (double)beta1),
                       (float)(1.0 - (double)beta2), (float)((double)lr / bc1), (float)sqrt(bc2), step == 1 ? 1 : 0,
                       (uint16_t*)bf16_mirror);
    return yv_launch_status();
}

extern "C" int yv_axpby(float* dst, const float* src, size_t n, float a, float b, void* stream) {
    if (!dst || !src) return YV_ERR_ARG;
    if (n == 0) return YV_OK;
    const size_t want = (n + 255) / 256;
    hipLaunchKernelGGL(axpby_kernel, dim3((int)(want > 4096 ? 4096 : want)), dim3(256), 0, (hipStream_t)stream, dst, src, n, a, b);
    return yv_launch_status();
}

extern "C" int yv_ema_update(float* ema, const float* src, size_t n, float decay, void* stream) {
    return yv_axpby(ema, src, n, decay, (float)(1.0 - (double)decay), stream);
}

extern "C" int yv_loss_fwd_bwd(const float* logits, const int32_t* labels, int B, int nc, float w_lsce, float w_focal,
                               float* loss, float* grad, void* stream) {
    if (!logits || !labels || !loss || !grad || B <= 0 || nc <= 0) return YV_ERR_ARG;
    if (nc > LOSS_MAX_NC) return YV_ERR_LIMIT;
    hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, B, nc, w_lsce, w_focal,
                       loss, grad);
    return yv_launch_status();
}

extern "C" int yv_sgd_step(float* p, const float* g, float* m, size_t n, float lr, float momentum,
                           float weight_decay, float grad_scale, int first, void* bf16_mirror, void* stream) {
    if (!p || !g || !m) return YV_ERR_ARG;
    if (n == 0) return YV_OK;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m) & 15) return YV_ERR_ARG;
    size_t n4 = n >> 2;
    size_t want = (n4 + 255) / 256;
    int blocks = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    hipLaunchKernelGGL(sgd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, n, lr, momentum,
                       weight_decay, grad_scale, first, (uint16_t*)bf16_mirror);
    return yv_launch_status();
}

extern "C" int yv_transpose_bf16_batched(const void* src, void* dst, int rows, int cols, int batch, long long src_stride,
                                 long long dst_stride, void* stream) {
    if (!src || !dst || rows <= 0 || cols <= 0 || batch <= 0 || (rows & 7) || (cols & 7)) return YV_ERR_ARG;
    if (((uintptr_t)src & 15) || ((uintptr_t)dst & 15) || (src_stride & 7) || (dst_stride & 7)) return YV_ERR_ARG;
    if (batch > 65535 || (rows + 63) / 64 > 65535) return YV_ERR_LIMIT;
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((cols + 63) / 64, (rows + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)src, (uint16_t*)dst, rows, cols, src_stride, dst_stride);
    return yv_launch_status();
}
