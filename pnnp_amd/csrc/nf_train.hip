// NoiseFlow NLL fitting on the device: one [pre-scale, Conv2d1x1, AffineCoupling] pair of the forward chain in TRAINING mode
// (BatchNorm with batch statistics) and its backward pass.  Reference: archs/noise_flow.py:113-165 (forward/_loss/loss),
// trainer_NF_SID.py:102,116-126 (net.train(); loss(); backward()), flow_layers/affine_coupling.py:36-53,245-295,
// conv2d1x1.py:47-92, signal_dependant.py:37-70, gain.py:79-110; the backward is what autograd derives for those lines.
//
// A pair maps x [B][4][H][W] to z:
//   x' = x / sqrt(a*clean + b)        SignalDependantISO, first pair only (a = beta1/gain, b = beta2; device scalars)
//   v  = Wm x'                        Conv2d1x1 (W = P L U; the scalar GainISO division is folded into Wm by the caller)
//   h1 = conv3x3(v[0:2]) + b1;  a1 = relu(BN1(h1));  h2 = W2 a1 + b2;  a2 = relu(BN2(h2))
//   raw3 = conv3x3_valid(pad1([a2, ring]));  out = raw3 * exp(3 logs);  shift = out[0:2];  ls = scale * tanh(out[2:4])
//   z = [v0, v1, v2*exp(ls_a) + shift_a, v3*exp(ls_b) + shift_b];   log-det pixel terms: ls_a + ls_b - sum_c log sqrt(.)
// BatchNorm needs the statistics of h1 and h2 over the whole batch, so the forward is three streaming passes (conv1 ->
// conv2 -> couple) with a one-workgroup statistics kernel after the first two; the backward mirrors it (couple -> conv2 ->
// conv1) with the BatchNorm-backward sums reduced between passes.  Every pass reads/writes 4-plane fp32 maps: HBM-bound.
// Parameter sums are reduced deterministically: per-workgroup partial rows, then a column reduction in double.
// h1 and h2 are kept WITHOUT their conv biases: a training-mode BatchNorm subtracts the batch mean, so the bias cancels
// exactly, and leaving it out keeps (h - mean) free of the cancellation that (h + b) - (mean + b) suffers in fp32 when
// |b| >> std(h) (after GainISO the activations are ~1e-5 while biases are ~0.1: the ReLU masks would otherwise carry
// ~1e-3 relative noise).  bn mean1/mean2 are therefore the bias-free means; the caller adds b1/b2 for running_mean.
//
// Coupling parameter block `prm` (device, 301 floats; the gradient block has the same layout):
//   W1[4][2][9] @0, B1[4] @72, G1[4] @76 (BN1 weight), BE1[4] @80, W2[4][4] @84, B2[4] @100, G2[4] @104, BE2[4] @108,
//   W3[4][5][9] @112, B3[4] @292, LOGS[4] @296, SCALE @300
// Statistics block `bn` (device, 24 floats): mean1[4], rstd1[4], var1[4] (biased), mean2[4], rstd2[4], var2[4].
#include "common.h"

namespace {

constexpr int TS = 32, HS = TS + 2;            // output tile, tile + halo 1
[[maybe_unused]] constexpr int P_W1 = 0, P_B1 = 72, P_G1 = 76, P_BE1 = 80, P_W2 = 84, P_B2 = 100, P_G2 = 104, P_BE2 = 108, P_W3 = 112, P_B3 = 292,
              P_LOGS = 296, P_SCALE = 300;
constexpr float BN_EPS = 1e-5f;

// Sum over the 64 lanes with DPP moves (VALU rate, no LDS crossbar latency): after the four in-row steps every lane of a
// 16-lane row holds its row's sum; row_bcast15 / row_bcast31 then carry the sums upwards, leaving the total in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v = dpp_add<0xB1, 0xf>(v);       // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);       // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);      // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);      // row_mirror
    v = dpp_add<0x142, 0xa>(v);      // row_bcast15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);      // row_bcast31 into rows 2 and 3
    return v;
}

// sums N per-thread values over the 256-thread workgroup into out[0..N) (out: global row of the partial matrix)
template <int N>
__device__ __forceinline__ void block_sum_store(const float (&v)[N], float* __restrict__ out, float (*red)[4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float s = wave_sum_lane63(v[i]);
        if (lane == 63) red[i][wave] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 256) out[i] = (red[i][0] + red[i][1]) + (red[i][2] + red[i][3]);
}

struct PairIn {
    const float* x;          // [B][4][H][W]
    const float* clean;      // [B][4][H][W] or null
    const float* ab;         // device {a, b} of the signal-dependent scale (used iff clean)
    const float* wm;         // device [4][4]
    const float* prm;        // device [301]
    const float* bn;         // device [24]
    int H, W;
};

// x' = x / sqrt(a*clean + b) (or x) and the scale at one in-image pixel
__device__ __forceinline__ void load_xp(const PairIn& p, const float* xb, const float* cb, int64_t plane, int64_t pix, float (&xp)[4],
                                        float (&sc)[4]) {
    const float a = cb ? p.ab[0] : 0.f, b = cb ? p.ab[1] : 1.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        sc[c] = cb ? sqrtf(a * cb[c * plane + pix] + b) : 1.f;
        xp[c] = cb ? xb[c * plane + pix] / sc[c] : xb[c * plane + pix];
    }
}

// ---------------------------------------------------------------------------------------------------------- forward
// pass 1: h1 = conv2d_1(v[0:2]) and the per-workgroup sums of h1, h1^2
__global__ void __launch_bounds__(256)
nf_tr_conv1_kernel(PairIn p, float* __restrict__ h1, float* __restrict__ part) {
    __shared__ float vs[2][HS][HS + 1];
    __shared__ float red[8][4];
    const int b = blockIdx.z, ty0 = blockIdx.y * TS, tx0 = blockIdx.x * TS, H = p.H, W = p.W;
    const int64_t plane = (int64_t)H * W;
    const float* xb = p.x + (int64_t)b * 4 * plane;
    const float* cb = p.clean ? p.clean + (int64_t)b * 4 * plane : nullptr;
    for (int i = threadIdx.x; i < HS * HS; i += 256) {
        const int r = i / HS, q = i % HS, gy = ty0 + r - 1, gx = tx0 + q - 1;
        float v0 = 0.f, v1 = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            float xp[4], sc[4];
            load_xp(p, xb, cb, plane, (int64_t)gy * W + gx, xp, sc);
#pragma unroll
            for (int c = 0; c < 4; ++c) { v0 += p.wm[c] * xp[c]; v1 += p.wm[4 + c] * xp[c]; }
        }
        vs[0][r][q] = v0; vs[1][r][q] = v1;
    }
    __syncthreads();
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < TS * TS; i += 256) {
        const int r = i / TS, q = i % TS, gy = ty0 + r, gx = tx0 + q;
        if (gy >= H || gx >= W) continue;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float s = 0.f;                   // bias-free: see the note on h1/h2 at the top of the file
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < 9; ++t) s += p.prm[P_W1 + (o * 2 + c) * 9 + t] * vs[c][r + t / 3][q + t % 3];
            h1[((int64_t)b * 4 + o) * plane + (int64_t)gy * W + gx] = s;
            acc[o] += s; acc[4 + o] += s * s;
        }
    }
    block_sum_store<8>(acc, part + (((int64_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8, red);
}

// statistics of one BatchNorm: part [rows][8] (sum[4], sumsq[4]) -> bn[0:12] = mean, rstd, biased var
__global__ void __launch_bounds__(1024)
nf_tr_bnstat_kernel(const float* __restrict__ part, int rows, double inv_n, float* __restrict__ bn) {
    __shared__ double red[1024];
    const int col = threadIdx.x & 7;
    double s = 0.0;
    for (int r = threadIdx.x >> 3; r < rows; r += 128) s += (double)part[(int64_t)r * 8 + col];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 512; k >= 8; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x < 4) {
        const double m = red[threadIdx.x] * inv_n;
        double var = red[4 + threadIdx.x] * inv_n - m * m;
        if (var < 0.0) var = 0.0;
        bn[threadIdx.x] = (float)m;
        bn[4 + threadIdx.x] = (float)(1.0 / sqrt(var + (double)BN_EPS));
        bn[8 + threadIdx.x] = (float)var;
    }
}

// pass 2 (pointwise over B*H*W pixels, 1024 per workgroup): h2 = conv2d_2(relu(BN1(h1))) and the sums of h2, h2^2
__global__ void __launch_bounds__(256)
nf_tr_conv2_kernel(const float* __restrict__ h1, const float* __restrict__ prm, const float* __restrict__ bn, float* __restrict__ h2,
                   float* __restrict__ part, int64_t plane, int64_t npix) {
    __shared__ float red[8][4];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 4; ++k) {
        const int64_t g = (int64_t)blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (g >= npix) break;
        const int64_t b = g / plane, pix = g - b * plane, base = b * 4 * plane + pix;
        float a1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            a1[c] = fmaxf(prm[P_G1 + c] * ((h1[base + c * plane] - bn[c]) * bn[4 + c]) + prm[P_BE1 + c], 0.f);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) s += prm[P_W2 + o * 4 + c] * a1[c];
            h2[base + o * plane] = s;
            acc[o] += s; acc[4 + o] += s * s;
        }
    }
    block_sum_store<8>(acc, part + (int64_t)blockIdx.x * 8, red);
}

// a2 = relu(BN2(h2)) on tile + halo 1 (zero outside the image: the ConstantPad3d ring)
__device__ __forceinline__ void stage_a2(const PairIn& p, const float* __restrict__ h2b, float (*a2s)[HS][HS + 1], int ty0, int tx0) {
    const int64_t plane = (int64_t)p.H * p.W;
    for (int i = threadIdx.x; i < HS * HS; i += 256) {
        const int r = i / HS, q = i % HS, gy = ty0 + r - 1, gx = tx0 + q - 1;
        const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            a2s[c][r][q] = in ? fmaxf(p.prm[P_G2 + c] * ((h2b[c * plane + (int64_t)gy * p.W + gx] - p.bn[12 + c]) * p.bn[16 + c]) +
                                          p.prm[P_BE2 + c], 0.f)
                              : 0.f;
    }
}

// pass 3: conv2d_3 on [a2, ring], the affine coupling, the pixel log-det terms and sum z^2 per workgroup
__global__ void __launch_bounds__(256)
nf_tr_couple_kernel(PairIn p, const float* __restrict__ h2, float* __restrict__ z, float* __restrict__ out3, float* __restrict__ ldpart) {
    __shared__ float a2s[4][HS][HS + 1];
    __shared__ float red[2][4];
    const int b = blockIdx.z, ty0 = blockIdx.y * TS, tx0 = blockIdx.x * TS, H = p.H, W = p.W;
    const int64_t plane = (int64_t)H * W;
    const float* xb = p.x + (int64_t)b * 4 * plane;
    const float* cb = p.clean ? p.clean + (int64_t)b * 4 * plane : nullptr;
    stage_a2(p, h2 + (int64_t)b * 4 * plane, a2s, ty0, tx0);
    __syncthreads();
    float acc[2] = {0.f, 0.f};
    const float scale = p.prm[P_SCALE];
    for (int i = threadIdx.x; i < TS * TS; i += 256) {
        const int r = i / TS, q = i % TS, gy = ty0 + r, gx = tx0 + q;
        if (gy >= H || gx >= W) continue;
        const int64_t pix = (int64_t)gy * W + gx;
        float o3[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float s = p.prm[P_B3 + o];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = gy + t / 3 - 1, xx = gx + t % 3 - 1;
#pragma unroll
                for (int c = 0; c < 4; ++c) s += p.prm[P_W3 + (o * 5 + c) * 9 + t] * a2s[c][r + t / 3][q + t % 3];
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) s += p.prm[P_W3 + (o * 5 + 4) * 9 + t];
            }
            o3[o] = s * expf(3.f * p.prm[P_LOGS + o]);
            out3[((int64_t)b * 4 + o) * plane + pix] = o3[o];
        }
        float xp[4], sc[4], v[4] = {0.f, 0.f, 0.f, 0.f};
        load_xp(p, xb, cb, plane, pix, xp, sc);
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[o] += p.wm[o * 4 + c] * xp[c];
        const float lsa = scale * tanhf(o3[2]), lsb = scale * tanhf(o3[3]);
        const float z2 = v[2] * expf(lsa) + o3[0], z3 = v[3] * expf(lsb) + o3[1];
        float* zb = z + (int64_t)b * 4 * plane + pix;
        zb[0] = v[0]; zb[plane] = v[1]; zb[2 * plane] = z2; zb[3 * plane] = z3;
        float ld = lsa + lsb;
        if (cb) ld -= (logf(sc[0]) + logf(sc[1])) + (logf(sc[2]) + logf(sc[3]));
        acc[0] += ld;
        acc[1] += (v[0] * v[0] + v[1] * v[1]) + (z2 * z2 + z3 * z3);
    }
    block_sum_store<2>(acc, ldpart + (((int64_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 2, red);
}

// ---------------------------------------------------------------------------------------------------------- backward
// deterministic column sums: part [rows][cols] -> out[cols] (double accumulation), one workgroup per column
__global__ void __launch_bounds__(256)
nf_tr_colsum_kernel(const float* __restrict__ part, int rows, int cols, float* __restrict__ out) {
    __shared__ double red[256];
    const int col = blockIdx.x;
    double s = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) s += (double)part[(int64_t)r * cols + col];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[col] = (float)red[0];
}

constexpr int NB1 = 197;     // dW3[180] db3[4] dlogs[4] dscale[1] sum dy2[4] sum dy2*xhat2[4]
constexpr int NB2 = 28;      // dW2[16] db2[4] sum dy1[4] sum dy1*xhat1[4]
constexpr int NB3 = 94;      // dW1[72] db1[4] dWm[16] da db

// backward pass 1: through the coupling and conv2d_3 down to dy2 = dL/d(BN2 output) (ReLU mask applied).
//   dz: gradient of the pair's output (times dzmul; the last pair passes z itself and dzmul = g/(B*D): d(-log N(z))/dz = z)
//   cobj = dL/d(objective_b), the same for every crop.   dv23 [B][2][H][W]: gradient of v[2:4].
__global__ void __launch_bounds__(256)
nf_tr_bwd_couple_kernel(PairIn p, const float* __restrict__ h2, const float* __restrict__ out3, const float* __restrict__ dz, float dzmul,
                        float cobj, float* __restrict__ dy2, float* __restrict__ dv23, float* __restrict__ part) {
    __shared__ float a2s[4][HS][HS + 1];
    __shared__ float gs[4][HS][HS + 1];        // d raw3 on tile + halo 1 (zero outside the image)
    __shared__ float red[48][4];
    const int b = blockIdx.z, ty0 = blockIdx.y * TS, tx0 = blockIdx.x * TS, H = p.H, W = p.W;
    const int64_t plane = (int64_t)H * W;
    const float* xb = p.x + (int64_t)b * 4 * plane;
    const float* cb = p.clean ? p.clean + (int64_t)b * 4 * plane : nullptr;
    const float* h2b = h2 + (int64_t)b * 4 * plane;
    const float* ob = out3 + (int64_t)b * 4 * plane;
    const float* dzb = dz + (int64_t)b * 4 * plane;
    float* row = part + (((int64_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NB1;
    stage_a2(p, h2b, a2s, ty0, tx0);
    const float scale = p.prm[P_SCALE];
    float sm[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // db3[4], dlogs[4], dscale
    for (int i = threadIdx.x; i < HS * HS; i += 256) {
        const int r = i / HS, q = i % HS, gy = ty0 + r - 1, gx = tx0 + q - 1;
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const int64_t pix = (int64_t)gy * W + gx;
            float xp[4], sc[4], v2 = 0.f, v3 = 0.f;
            load_xp(p, xb, cb, plane, pix, xp, sc);
#pragma unroll
            for (int c = 0; c < 4; ++c) { v2 += p.wm[8 + c] * xp[c]; v3 += p.wm[12 + c] * xp[c]; }
            const float o0 = ob[pix], o1 = ob[plane + pix], o2 = ob[2 * plane + pix], o3 = ob[3 * plane + pix];
            const float dza = dzb[2 * plane + pix] * dzmul, dzb_ = dzb[3 * plane + pix] * dzmul;
            const float ta = tanhf(o2), tb = tanhf(o3), ea = expf(scale * ta), eb = expf(scale * tb);
            const float dlsa = dza * v2 * ea + cobj, dlsb = dzb_ * v3 * eb + cobj;
            const float dout[4] = {dza, dzb_, dlsa * scale * (1.f - ta * ta), dlsb * scale * (1.f - tb * tb)};
            const float outv[4] = {o0, o1, o2, o3};
#pragma unroll
            for (int o = 0; o < 4; ++o) g[o] = dout[o] * expf(3.f * p.prm[P_LOGS + o]);
            const bool interior = r >= 1 && r <= TS && q >= 1 && q <= TS;        // this workgroup owns the pixel
            if (interior) {
#pragma unroll
                for (int o = 0; o < 4; ++o) { sm[o] += g[o]; sm[4 + o] += 3.f * dout[o] * outv[o]; }
                sm[8] += dlsa * ta + dlsb * tb;
                dv23[((int64_t)b * 2) * plane + pix] = dza * ea;
                dv23[((int64_t)b * 2 + 1) * plane + pix] = dzb_ * eb;
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) gs[o][r][q] = g[o];
    }
    __syncthreads();
    // d a2 = conv3^T(g), ReLU mask, BatchNorm-backward sums
    float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < TS * TS; i += 256) {
        const int r = i / TS, q = i % TS, gy = ty0 + r, gx = tx0 + q;
        if (gy >= H || gx >= W) continue;
        const int64_t pix = (int64_t)gy * W + gx;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float s = 0.f;
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int t = 0; t < 9; ++t)      // output pixel (p - (t - centre)) used a2(p) through tap t
                    s += p.prm[P_W3 + (o * 5 + c) * 9 + t] * gs[o][r + 2 - t / 3][q + 2 - t % 3];
            const float hv = h2b[c * plane + pix];
            const float xh = (hv - p.bn[12 + c]) * p.bn[16 + c];
            const float pre = p.prm[P_G2 + c] * xh + p.prm[P_BE2 + c];
            const float d = pre > 0.f ? s : 0.f;
            dy2[((int64_t)b * 4 + c) * plane + pix] = d;
            bs[c] += d; bs[4 + c] += d * xh;
        }
    }
    // dW3[o][c][t] = sum_p g[o](p) * pad(a2)[c](p + t): one output channel at a time (45 accumulators; two at a time
    // costs more in occupancy than it saves in LDS reads: measured 71 vs 61 us)
#pragma unroll 1
    for (int o = 0; o < 4; ++o) {
        float acc[45];
#pragma unroll
        for (int j = 0; j < 45; ++j) acc[j] = 0.f;
        for (int i = threadIdx.x; i < TS * TS; i += 256) {
            const int r = i / TS, q = i % TS, gy = ty0 + r, gx = tx0 + q;
            if (gy >= H || gx >= W) continue;
            const float gv = gs[o][r + 1][q + 1];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = gy + t / 3 - 1, xx = gx + t % 3 - 1;
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c * 9 + t] += gv * a2s[c][r + t / 3][q + t % 3];
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) acc[36 + t] += gv;
            }
        }
        block_sum_store<45>(acc, row + o * 45, red);
    }
    block_sum_store<9>(sm, row + 180, red);
    block_sum_store<8>(bs, row + 189, red);
}

// backward pass 2 (pointwise): BatchNorm2 backward, conv2d_2 backward, ReLU mask of layer 1.
//   s2 [8] = reduced (sum dy2[4], sum dy2*xhat2[4]).   Writes dy1 = dL/d(BN1 output).
__global__ void __launch_bounds__(256)
nf_tr_bwd_conv2_kernel(const float* __restrict__ h1, const float* __restrict__ h2, const float* __restrict__ dy2, const float* __restrict__ prm,
                       const float* __restrict__ bn, const float* __restrict__ s2, float inv_n, float* __restrict__ dy1,
                       float* __restrict__ part, int64_t plane, int64_t npix) {
    __shared__ float red[NB2][4];
    float acc[NB2];
#pragma unroll
    for (int j = 0; j < NB2; ++j) acc[j] = 0.f;
    for (int k = 0; k < 4; ++k) {
        const int64_t g = (int64_t)blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (g >= npix) break;
        const int64_t b = g / plane, pix = g - b * plane, base = b * 4 * plane + pix;
        float a1[4], pre1[4], xh1[4], dh2[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            xh1[c] = (h1[base + c * plane] - bn[c]) * bn[4 + c];
            pre1[c] = prm[P_G1 + c] * xh1[c] + prm[P_BE1 + c];
            a1[c] = fmaxf(pre1[c], 0.f);
            const float xh2 = (h2[base + c * plane] - bn[12 + c]) * bn[16 + c];
            dh2[c] = prm[P_G2 + c] * bn[16 + c] * (dy2[base + c * plane] - s2[c] * inv_n - xh2 * (s2[4 + c] * inv_n));
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[o * 4 + c] += dh2[o] * a1[c];
            acc[16 + o] += dh2[o];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float s = 0.f;
#pragma unroll
            for (int o = 0; o < 4; ++o) s += prm[P_W2 + o * 4 + c] * dh2[o];
            const float d = pre1[c] > 0.f ? s : 0.f;
            dy1[base + c * plane] = d;
            acc[20 + c] += d; acc[24 + c] += d * xh1[c];
        }
    }
    block_sum_store<NB2>(acc, part + (int64_t)blockIdx.x * NB2, red);
}

// backward pass 3: BatchNorm1 backward, conv2d_1 backward, the Conv2d1x1 / pre-scale backward; writes dx.
//   s1 [8] = reduced (sum dy1[4], sum dy1*xhat1[4]);  dz planes 0,1 (times dzmul) and dv23 give the gradient of v.
__global__ void __launch_bounds__(256)
nf_tr_bwd_conv1_kernel(PairIn p, const float* __restrict__ h1, const float* __restrict__ dy1, const float* __restrict__ s1, float inv_n,
                       const float* __restrict__ dz, float dzmul, const float* __restrict__ dv23, float cobj, float* __restrict__ dx,
                       float* __restrict__ part) {
    __shared__ float vs[2][HS][HS + 1];
    __shared__ float ds[4][HS][HS + 1];        // d h1 on tile + halo 1 (zero outside the image)
    __shared__ float red[NB3][4];
    const int b = blockIdx.z, ty0 = blockIdx.y * TS, tx0 = blockIdx.x * TS, H = p.H, W = p.W;
    const int64_t plane = (int64_t)H * W;
    const float* xb = p.x + (int64_t)b * 4 * plane;
    const float* cb = p.clean ? p.clean + (int64_t)b * 4 * plane : nullptr;
    const float* h1b = h1 + (int64_t)b * 4 * plane;
    const float* dyb = dy1 + (int64_t)b * 4 * plane;
    for (int i = threadIdx.x; i < HS * HS; i += 256) {
        const int r = i / HS, q = i % HS, gy = ty0 + r - 1, gx = tx0 + q - 1;
        float v0 = 0.f, v1 = 0.f, d[4] = {0.f, 0.f, 0.f, 0.f};
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const int64_t pix = (int64_t)gy * W + gx;
            float xp[4], sc[4];
            load_xp(p, xb, cb, plane, pix, xp, sc);
#pragma unroll
            for (int c = 0; c < 4; ++c) { v0 += p.wm[c] * xp[c]; v1 += p.wm[4 + c] * xp[c]; }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float xh = (h1b[c * plane + pix] - p.bn[c]) * p.bn[4 + c];
                d[c] = p.prm[P_G1 + c] * p.bn[4 + c] * (dyb[c * plane + pix] - s1[c] * inv_n - xh * (s1[4 + c] * inv_n));
            }
        }
        vs[0][r][q] = v0; vs[1][r][q] = v1;
#pragma unroll
        for (int c = 0; c < 4; ++c) ds[c][r][q] = d[c];
    }
    __syncthreads();
    float acc[NB3];
#pragma unroll
    for (int j = 0; j < NB3; ++j) acc[j] = 0.f;
    for (int i = threadIdx.x; i < TS * TS; i += 256) {
        const int r = i / TS, q = i % TS, gy = ty0 + r, gx = tx0 + q;
        if (gy >= H || gx >= W) continue;
        const int64_t pix = (int64_t)gy * W + gx;
        float dv[4];
        dv[0] = dz[((int64_t)b * 4) * plane + pix] * dzmul;
        dv[1] = dz[((int64_t)b * 4 + 1) * plane + pix] * dzmul;
        dv[2] = dv23[((int64_t)b * 2) * plane + pix];
        dv[3] = dv23[((int64_t)b * 2 + 1) * plane + pix];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const float dc = ds[o][r + 1][q + 1];
            acc[72 + o] += dc;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    acc[(o * 2 + c) * 9 + t] += dc * vs[c][r + t / 3][q + t % 3];
                    dv[c] += p.prm[P_W1 + (o * 2 + c) * 9 + t] * ds[o][r + 2 - t / 3][q + 2 - t % 3];
                }
        }
        float xp[4], sc[4];
        load_xp(p, xb, cb, plane, pix, xp, sc);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float dxp = 0.f;
#pragma unroll
            for (int o = 0; o < 4; ++o) { dxp += p.wm[o * 4 + j] * dv[o]; acc[76 + o * 4 + j] += dv[o] * xp[j]; }
            if (cb) {          // x' = x / sc, objective -= log sc, sc = sqrt(a*clean + b)
                const float dsc = -(dxp * xp[j] + cobj) / sc[j];
                const float ds2 = dsc / (2.f * sc[j]);
                acc[92] += ds2 * cb[j * plane + pix]; acc[93] += ds2;
                dxp = dxp / sc[j];
            }
            dx[((int64_t)b * 4 + j) * plane + pix] = dxp;
        }
    }
    block_sum_store<NB3>(acc, part + (((int64_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NB3, red);
}

inline bool bad_shape(int B, int H, int W) { return B <= 0 || H <= 0 || W <= 0; }
inline dim3 tile_grid(int B, int H, int W) { return dim3((W + TS - 1) / TS, (H + TS - 1) / TS, B); }

}  // namespace

extern "C" {

int pnnp_nf_train_tiles(int B, int H, int W) { return bad_shape(B, H, W) ? 0 : B * ((H + TS - 1) / TS) * ((W + TS - 1) / TS); }
int pnnp_nf_train_pblocks(int B, int H, int W) { return bad_shape(B, H, W) ? 0 : (int)(((int64_t)B * H * W + 1023) / 1024); }

// Forward of one pair in training mode.  Saves h1, h2, out3 ([B][4][H][W] each) for the backward; bn [24] receives the
// batch statistics; ldpart [tiles][2] the per-workgroup (log-det pixel terms, sum z^2); part: scratch of
// max(tiles, pblocks) * 8 floats.  clean/ab: SignalDependantISO (first pair) or null.  All pointers are device pointers.
int pnnp_nf_train_fwd_pair_f32(const float* x, const float* clean, const float* ab, const float* wm, const float* prm, float* bn,
                               float* h1, float* h2, float* out3, float* z, float* ldpart, float* part, int B, int H, int W,
                               void* stream) {
    if (bad_shape(B, H, W) || !x || !wm || !prm || !bn || !h1 || !h2 || !out3 || !z || !ldpart || !part || (clean && !ab) || x == z)
        return PNNP_E_INVALID;
    hipStream_t st = as_stream(stream);
    const PairIn p{x, clean, ab, wm, prm, bn, H, W};
    const dim3 grid = tile_grid(B, H, W);
    const int tiles = pnnp_nf_train_tiles(B, H, W), pb = pnnp_nf_train_pblocks(B, H, W);
    const int64_t plane = (int64_t)H * W, npix = (int64_t)B * plane;
    const double inv_n = 1.0 / (double)npix;
    hipLaunchKernelGGL(nf_tr_conv1_kernel, grid, dim3(256), 0, st, p, h1, part);
    hipLaunchKernelGGL(nf_tr_bnstat_kernel, dim3(1), dim3(1024), 0, st, part, tiles, inv_n, bn);
    hipLaunchKernelGGL(nf_tr_conv2_kernel, dim3(pb), dim3(256), 0, st, h1, prm, bn, h2, part, plane, npix);
    hipLaunchKernelGGL(nf_tr_bnstat_kernel, dim3(1), dim3(1024), 0, st, part, pb, inv_n, bn + 12);
    hipLaunchKernelGGL(nf_tr_couple_kernel, grid, dim3(256), 0, st, p, h2, z, out3, ldpart);
    return pnnp_launch_status();
}

// Batch statistics of a coupling's two BatchNorm layers for a tensor u whose first two planes feed the coupling network directly
// (training-mode SAMPLING, trainer_LRID.py:34-39,420-427: the proxy is never put in eval mode): h1 = conv2d_1(u[0:2]) -> bn[0:12],
// h2 = conv2d_2(relu(BN1(h1))) -> bn[12:24].  `ident`: device [16] floats, the 4x4 identity (the Conv2d1x1 of the pair acts AFTER
// the coupling in this direction).  h1, h2: scratch [B][4][H][W]; part: max(tiles, pblocks) * 8 floats.
int pnnp_nf_train_stats_f32(const float* u, const float* ident, const float* prm, float* bn, float* h1, float* h2, float* part,
                            int B, int H, int W, void* stream) {
    if (bad_shape(B, H, W) || !u || !ident || !prm || !bn || !h1 || !h2 || !part) return PNNP_E_INVALID;
    hipStream_t st = as_stream(stream);
    const PairIn p{u, nullptr, nullptr, ident, prm, bn, H, W};
    const dim3 grid = tile_grid(B, H, W);
    const int tiles = pnnp_nf_train_tiles(B, H, W), pb = pnnp_nf_train_pblocks(B, H, W);
    const int64_t plane = (int64_t)H * W, npix = (int64_t)B * plane;
    const double inv_n = 1.0 / (double)npix;
    hipLaunchKernelGGL(nf_tr_conv1_kernel, grid, dim3(256), 0, st, p, h1, part);
    hipLaunchKernelGGL(nf_tr_bnstat_kernel, dim3(1), dim3(1024), 0, st, part, tiles, inv_n, bn);
    hipLaunchKernelGGL(nf_tr_conv2_kernel, dim3(pb), dim3(256), 0, st, h1, prm, bn, h2, part, plane, npix);
    hipLaunchKernelGGL(nf_tr_bnstat_kernel, dim3(1), dim3(1024), 0, st, part, pb, inv_n, bn + 12);
    return pnnp_launch_status();
}

// Backward of one pair.  dz: gradient of the pair's output (multiplied by dzmul); cobj = dL/d(objective).  Outputs: dx
// [B][4][H][W]; sums [319] = dW3[180] db3[4] dlogs[4] dscale[1] dBE2[4] dG2[4] | dW2[16] db2[4] dBE1[4] dG1[4] |
// dW1[72] db1[4] dWm[16] da db.  Scratch: dy [B][4][H][W] x2 (dy2, dy1), dv23 [B][2][H][W], part: max(tiles*197, pblocks*28) floats.
int pnnp_nf_train_bwd_pair_f32(const float* x, const float* clean, const float* ab, const float* wm, const float* prm, const float* bn,
                               const float* h1, const float* h2, const float* out3, const float* dz, float dzmul, float cobj,
                               float* dx, float* sums, float* dy2, float* dy1, float* dv23, float* part, int B, int H, int W,
                               void* stream) {
    if (bad_shape(B, H, W) || !x || !wm || !prm || !bn || !h1 || !h2 || !out3 || !dz || !dx || !sums || !dy2 || !dy1 || !dv23 || !part ||
        (clean && !ab) || dx == dz)
        return PNNP_E_INVALID;
    hipStream_t st = as_stream(stream);
    const PairIn p{x, clean, ab, wm, prm, bn, H, W};
    const dim3 grid = tile_grid(B, H, W);
    const int tiles = pnnp_nf_train_tiles(B, H, W), pb = pnnp_nf_train_pblocks(B, H, W);
    const int64_t plane = (int64_t)H * W, npix = (int64_t)B * plane;
    const float inv_n = (float)(1.0 / (double)npix);
    hipLaunchKernelGGL(nf_tr_bwd_couple_kernel, grid, dim3(256), 0, st, p, h2, out3, dz, dzmul, cobj, dy2, dv23, part);
    hipLaunchKernelGGL(nf_tr_colsum_kernel, dim3(NB1), dim3(256), 0, st, part, tiles, NB1, sums);
    hipLaunchKernelGGL(nf_tr_bwd_conv2_kernel, dim3(pb), dim3(256), 0, st, h1, h2, dy2, prm, bn, sums + 189, inv_n, dy1, part, plane, npix);
    hipLaunchKernelGGL(nf_tr_colsum_kernel, dim3(NB2), dim3(256), 0, st, part, pb, NB2, sums + NB1);
    hipLaunchKernelGGL(nf_tr_bwd_conv1_kernel, grid, dim3(256), 0, st, p, h1, dy1, sums + NB1 + 20, inv_n, dz, dzmul, dv23, cobj, dx, part);
    hipLaunchKernelGGL(nf_tr_colsum_kernel, dim3(NB3), dim3(256), 0, st, part, tiles, NB3, sums + NB1 + NB2);
    return pnnp_launch_status();
}

}  // extern "C"
