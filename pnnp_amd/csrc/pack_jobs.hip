// Weight re-packing as ONE launch per optimiser step.  The optimiser moves every weight every step, so every layer's packed
// images (direct forward / backward-data order, Winograd U = G g G^T forward / backward-data) are rebuilt each step: ~70
// launches of 4-8 us each when done layer by layer (0.38 ms of a 30 ms step).  Here a launch takes a table of jobs (a kernel
// argument, <= 32 per launch) and every workgroup finds its job from a prefix of block counts.
//   job kind 0: strided gather  dst[((t*(K/4)+kq)*Ndst + n_off + n)*4 + kr] = src[off + k*sk + n*sn + ts*st]  (k < Kvalid else 0)
//   job kind 1: Winograd filter transform into the kernel's chunked order (csrc/wino.hip)
//   job kind 2: 3x3 filters split into three bf16 pieces in the order csrc/conv_x3.hip streams them by LDS-DMA
//   job kind 3: a strided K x N matrix (1x1 / ConvTranspose / strided-tap weights) split likewise for csrc/gemm_x3.hip
//   job kind 4: 3x3 filters scaled by a power of two and split into two fp16 pieces for csrc/conv_h2s.hip (the scale comes from the amax slot
//               that a kind-5 job of an EARLIER launch filled)
//   job kind 5: max |src| -> atomicMax into a 4-byte slot (csrc/h2.h)
// The builders below produce exactly the jobs the per-layer entry points used to launch (same formulas, same layouts).
#include "common.h"
#include "h2.h"

namespace {

constexpr int W_KC = 8, W_BN = 64, W_UPLANE = W_BN * 4, W_US_STAGE = 16 * 2 * W_UPLANE;   // csrc/wino.hip (static_assert there)
constexpr int MAXJ = 32;

struct JobTable { int n; int blk_end[MAXJ]; PnnpPackJob job[MAXJ]; };

__device__ __forceinline__ void gather_job(const PnnpPackJob& j, int64_t blk, int nblk) {
    const float* __restrict__ src = j.src; float* __restrict__ dst = j.dst;
    const int T = j.T, K = j.K, N = j.N;
    const int64_t total = (int64_t)T * K * N;
    for (int64_t i = blk * 256 + threadIdx.x; i < total; i += (int64_t)nblk * 256) {
        const int kr = (int)(i & 3);
        int64_t r = i >> 2;
        const int n = (int)(r % N); r /= N;
        const int kq = (int)(r % (K >> 2));
        const int t = (int)(r / (K >> 2));
        const int k = kq * 4 + kr;
        const int ts = j.flip ? T - 1 - t : t;
        dst[(((int64_t)t * (K >> 2) + kq) * j.Ndst + j.n_off + n) * 4 + kr] = k < j.Kvalid ? src[j.off + k * j.sk + n * j.sn + ts * j.st] : 0.f;
    }
}

// U = G g G^T of every (k, n) filter, written in the Winograd kernel's chunked order [N/64][K/8][16][2][64][4].
//   forward:  g[a][b] = w[n][k][a][b]        (w [Cout][Cin][3][3], K = Cin, N = Cout)
//   dgrad:    g[a][b] = w[k][n][2-a][2-b]    (K = Cout, N = Cin)
__device__ __forceinline__ void wino_job(const PnnpPackJob& j, int64_t blk, int nblk) {
    const float* __restrict__ w = j.src; float* __restrict__ u = j.dst;
    const int Cout = j.K, Cin = j.N, dgrad = j.T;
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    const int64_t total = (int64_t)K * N;
    for (int64_t t = blk * 256 + threadIdx.x; t < total; t += (int64_t)nblk * 256) {
        const int n = (int)(t % N), k = (int)(t / N);
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                g[r][c] = dgrad ? w[((int64_t)k * Cin + n) * 9 + (2 - r) * 3 + (2 - c)] : w[((int64_t)n * Cin + k) * 9 + r * 3 + c];
        float gg[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            gg[0][c] = g[0][c];
            gg[1][c] = 0.5f * (g[0][c] + g[1][c] + g[2][c]);
            gg[2][c] = 0.5f * (g[0][c] - g[1][c] + g[2][c]);
            gg[3][c] = g[2][c];
        }
        const int nb = n / W_BN, nn = n % W_BN, kc = k / W_KC, kg = (k % W_KC) / 4, e = k % 4;
        float* o = u + ((int64_t)nb * (K / W_KC) + kc) * W_US_STAGE + (kg * W_BN + nn) * 4 + e;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float uu[4] = {gg[i][0], 0.5f * (gg[i][0] + gg[i][1] + gg[i][2]), 0.5f * (gg[i][0] - gg[i][1] + gg[i][2]), gg[i][2]};
#pragma unroll
            for (int q = 0; q < 4; ++q) o[(i * 4 + q) * 2 * W_UPLANE] = uu[q];
        }
    }
}

// bf16x3 pack of a 3x3 Conv2d weight for csrc/conv_x3.hip:  dst (uint16) [N/32][K16][tap 9][octet 2][piece 3][32][8]
//   element (k = chunk*16 + octet*8 + e, n = nb*32 + nn, tap):  forward  W = w[n][k][tap]        (K = Cin, N = Cout)
//                                                               dgrad    W = w[k][n][8 - tap]    (K = Cout, N = Cin)
//   pieces hi = bf16(W), mid = bf16(W - hi), lo = bf16(W - hi - mid) (round to nearest even; W = hi + mid + lo exactly);
//   k >= Kvalid (channel padding up to a multiple of 16) -> 0.   job: K = Cout, N = Cin, T = dgrad, Kvalid = padded K.
__device__ __forceinline__ unsigned short bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);      // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ void x3_job(const PnnpPackJob& j, int64_t blk, int nblk) {
    const float* __restrict__ w = j.src; unsigned short* __restrict__ u = reinterpret_cast<unsigned short*>(j.dst);
    const int Cout = j.K, Cin = j.N, dgrad = j.T;
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    const int Kp = j.Kvalid, K16 = Kp / 16;                       // padded reduction length
    const int64_t total = (int64_t)Kp * ((N + 31) / 32 * 32) * 9;
    for (int64_t t = blk * 256 + threadIdx.x; t < total; t += (int64_t)nblk * 256) {
        // enumerate in DESTINATION order (coalesced 2-byte stores of one piece plane; the three planes are 512 B apart)
        const int e = (int)(t & 7);
        int64_t r = t >> 3;
        const int nn = (int)(r & 31); r >>= 5;
        const int oct = (int)(r & 1); r >>= 1;
        const int tap = (int)(r % 9); r /= 9;
        const int c = (int)(r % K16);
        const int nb = (int)(r / K16);
        const int k = c * 16 + oct * 8 + e, n = nb * 32 + nn;
        float v = 0.f;
        if (k < K && n < N) v = dgrad ? w[((int64_t)k * Cin + n) * 9 + (8 - tap)] : w[((int64_t)n * Cin + k) * 9 + tap];
        const unsigned short h = bf16_rne(v);
        const float r1 = v - __uint_as_float((unsigned)h << 16);
        const unsigned short m = bf16_rne(r1);
        const float r2 = r1 - __uint_as_float((unsigned)m << 16);
        const unsigned short l = bf16_rne(r2);
        unsigned short* o = u + ((((int64_t)nb * K16 + c) * 9 + tap) * 2 + oct) * (3 * 32 * 8) + nn * 8 + e;
        o[0] = h; o[32 * 8] = m; o[2 * 32 * 8] = l;
    }
}

// bf16x3 pack of a K x N weight matrix for csrc/gemm_x3.hip:  dst (uint16) [Ntot/32][K16tot][octet 2][piece 3][32][8]
//   this job's sub-matrix: element (k, n), k < K, n < N  =  src[off + k*sk + n*sn], placed at row k_off + k (field `st`), column
//   n_off + n of the whole matrix (K16tot = field T, Ntot = field Ndst).  Rows / columns no job covers must be zero-filled by the
//   caller (the builders below cover everything).
__device__ __forceinline__ void x3mat_job(const PnnpPackJob& j, int64_t blk, int nblk) {
    const float* __restrict__ w = j.src; unsigned short* __restrict__ u = reinterpret_cast<unsigned short*>(j.dst);
    const int K = j.K, N = j.N, K16 = j.T, k_off = (int)j.st;
    const int64_t total = (int64_t)K * N;
    for (int64_t t = blk * 256 + threadIdx.x; t < total; t += (int64_t)nblk * 256) {
        const int kr = (int)(t & 7);                               // 8 consecutive k of one column: one 16-byte word per piece
        int64_t r = t >> 3;
        const int n = (int)(r % N);
        const int k = (int)(r / N) * 8 + kr;
        if (k >= K) continue;
        const float v = w[j.off + (int64_t)k * j.sk + (int64_t)n * j.sn];
        const unsigned short h = bf16_rne(v);
        const float r1 = v - __uint_as_float((unsigned)h << 16);
        const unsigned short m = bf16_rne(r1);
        const unsigned short l = bf16_rne(r1 - __uint_as_float((unsigned)m << 16));
        const int kk = k_off + k, nn = j.n_off + n;
        unsigned short* o = u + ((((int64_t)(nn >> 5) * K16 + (kk >> 4)) * 2 + ((kk >> 3) & 1)) * 3) * 256 + (nn & 31) * 8 + (kk & 7);
        o[0] = h; o[256] = m; o[512] = l;
    }
}

// fp16x2 pack of a K x N weight matrix for csrc/gemm_h2s.hip:  dst (fp16) [Ntot/32][K32tot][piece 2: hi', lo'][octet 4][32][8]  (4096 bytes per
//   32-channel item and 32-column block); sub-matrix addressing exactly as x3mat_job (K32tot = field T); W s with s = 2^pnnp_h2_scale_exp(*amax)
//   of the WHOLE weight tensor (a kind-5 job), hi' = f16(W s), lo' = f16(W s - hi').  Rows / columns no job covers: zero-filled by the caller.
__device__ __forceinline__ void h2mat_job(const PnnpPackJob& j, int64_t blk, int nblk) {
    const float* __restrict__ w = j.src; unsigned short* __restrict__ u = reinterpret_cast<unsigned short*>(j.dst);
    const int K = j.K, N = j.N, K32 = j.T, k_off = (int)j.st;
    const float s = __uint_as_float((unsigned)(pnnp_h2_scale_exp(j.amax[0]) + 127) << 23);
    const unsigned total = (unsigned)(K / 8) * (unsigned)N;          // one thread = 8 consecutive k of one column: a 16-byte word per piece (K % 8 == 0)
    for (unsigned t = (unsigned)blk * 256 + threadIdx.x; t < total; t += (unsigned)nblk * 256) {
        const unsigned n = t % (unsigned)N, k0 = (t / (unsigned)N) * 8;
        const float* src = w + j.off + (int64_t)k0 * j.sk + (int64_t)n * j.sn;
        unsigned hw[4], lw[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const float v0 = src[(int64_t)e * j.sk] * s, v1 = src[(int64_t)(e + 1) * j.sk] * s;
            const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
            const _Float16 l0 = (_Float16)(v0 - (float)h0), l1 = (_Float16)(v1 - (float)h1);
            hw[e >> 1] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            lw[e >> 1] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
        }
        const int kk = k_off + (int)k0, nn = j.n_off + (int)n;
        unsigned short* o = u + ((int64_t)(nn >> 5) * K32 + (kk >> 5)) * 2048 + ((kk >> 3) & 3) * 256 + (nn & 31) * 8;
        *reinterpret_cast<uint4*>(o) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        *reinterpret_cast<uint4*>(o + 1024) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
}

// fp16x2 pack of a 3x3 Conv2d weight for csrc/conv_h2s.hip:  dst (fp16) [N/32][K16] x { hi' [tap 9][octet 2][32][8], lo' [tap 10][octet 2][32][8] }
//   (the tenth lo' tap is ZERO: the partner of the unpaired ninth tap; 19456 bytes per 32-column block and chunk)
//   element (k, n, tap) as in x3_job;  W s with s = 2^pnnp_h2_scale_exp(*amax) (amax >= max |w|: a kind-5 job), hi' = f16(W s), lo' = f16(W s - hi')
//   (round to nearest even; the residual is exact in float32).  job: K = Cout, N = Cin, T = dgrad, Kvalid = padded K, amax = the slot.
__device__ __forceinline__ void h2_job(const PnnpPackJob& j, int64_t blk, int nblk) {
    const float* __restrict__ w = j.src; unsigned short* __restrict__ u = reinterpret_cast<unsigned short*>(j.dst);
    const int Cout = j.K, Cin = j.N, dgrad = j.T;
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    const int Kp = j.Kvalid, K16 = Kp / 16, NB = (N + 31) / 32;
    const float s = __uint_as_float((unsigned)(pnnp_h2_scale_exp(j.amax[0]) + 127) << 23);
    // one thread = one 16-byte word of each piece plane: 8 consecutive k of (32-column block nb, chunk c, tap, octet, column nn) -- 32-bit index
    // arithmetic, two 16-byte stores (was: one element, two 2-byte stores and three 64-bit divisions per thread: 45 us per launch)
    const unsigned words = (unsigned)NB * K16 * 9 * 2 * 32;        // (< 2^31 / 8: the launcher's job_blocks bound)
    for (unsigned t = (unsigned)blk * 256 + threadIdx.x; t < (unsigned)NB * K16 * 64; t += (unsigned)nblk * 256)        // the zero taps: 64 words per block
        reinterpret_cast<uint4*>(u)[(size_t)(t >> 6) * (9728 / 8) + 18 * 64 + (t & 63)] = make_uint4(0, 0, 0, 0);
    for (unsigned t = (unsigned)blk * 256 + threadIdx.x; t < words; t += (unsigned)nblk * 256) {
        const unsigned nn = t & 31, oct = (t >> 5) & 1;
        unsigned r = t >> 6;
        const unsigned tap = r % 9; r /= 9;
        const unsigned c = r % (unsigned)K16, nb = r / (unsigned)K16;
        const int k0 = (int)(c * 16 + oct * 8), n = (int)(nb * 32 + nn);
        unsigned hw[4], lw[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            float v[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = k0 + e + q;
                v[q] = 0.f;
                if (k < K && n < N) v[q] = dgrad ? w[((int64_t)k * Cin + n) * 9 + (8 - tap)] : w[((int64_t)n * Cin + k) * 9 + tap];
            }
            const float v0 = v[0] * s, v1 = v[1] * s;
            const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
            const _Float16 l0 = (_Float16)(v0 - (float)h0), l1 = (_Float16)(v1 - (float)h1);
            hw[e >> 1] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            lw[e >> 1] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
        }
        unsigned short* o = u + ((size_t)nb * K16 + c) * 9728 + tap * 512 + oct * 256 + nn * 8;
        *reinterpret_cast<uint4*>(o) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        *reinterpret_cast<uint4*>(o + 9 * 512) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
}

// max |src[0 .. n)| -> slot (non-negative floats order like their bit patterns; NaN counts as larger than inf: a diverged tensor is flagged)
__device__ __forceinline__ void amax_job(const PnnpPackJob& j, int64_t blk, int nblk) {
    const float* __restrict__ x = j.src;
    const int64_t n = j.sk;
    unsigned m = 0u;
    auto upd = [&](float v) { const unsigned b = __float_as_uint(v) & 0x7fffffffu; m = b > m ? b : m; };
    const int64_t n4 = (((uintptr_t)x) & 15) ? 0 : n >> 2;         // 16-byte loads where the tensor allows
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
    for (int64_t i = blk * 256 + threadIdx.x; i < n4; i += (int64_t)nblk * 256) { const float4 v = x4[i]; upd(v.x); upd(v.y); upd(v.z); upd(v.w); }
    for (int64_t i = n4 * 4 + blk * 256 + threadIdx.x; i < n; i += (int64_t)nblk * 256) upd(x[i]);
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)m, sft, 64); m = o > m ? o : m; }
    // ONE atomic per block (round 6: one per wave -- up to 4 096 on one slot for a 512 x 512 x 9 tensor, ~15 ns each and serialised: the launch that held
    // the deep layers' slots took 84 us, most of it this queue; csrc/common.h pnnp_amax_commit_block tells the same story for the activations)
    __shared__ unsigned amax_red[4];
    if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) m = amax_red[i] > m ? amax_red[i] : m;
        if (m) atomicMax(reinterpret_cast<unsigned*>(j.dst), m);
    }
}

__global__ void __launch_bounds__(256) pack_jobs_kernel(const JobTable tb) {
    int j = 0;
    while (j + 1 < tb.n && (int)blockIdx.x >= tb.blk_end[j]) ++j;          // uniform: <= 32 scalar compares
    const int b0 = j ? tb.blk_end[j - 1] : 0;
    const int nblk = tb.blk_end[j] - b0;
    const PnnpPackJob& job = tb.job[j];
    if (job.kind == 1) wino_job(job, (int64_t)blockIdx.x - b0, nblk);
    else if (job.kind == 2) x3_job(job, (int64_t)blockIdx.x - b0, nblk);
    else if (job.kind == 3) x3mat_job(job, (int64_t)blockIdx.x - b0, nblk);
    else if (job.kind == 4) h2_job(job, (int64_t)blockIdx.x - b0, nblk);
    else if (job.kind == 5) amax_job(job, (int64_t)blockIdx.x - b0, nblk);
    else if (job.kind == 6) h2mat_job(job, (int64_t)blockIdx.x - b0, nblk);
    else gather_job(job, (int64_t)blockIdx.x - b0, nblk);
}

int job_blocks(const PnnpPackJob& j) {
    if (j.kind == 5) { const int64_t b5 = (j.sk + 256 * 16 - 1) / (256 * 16); return (int)(b5 > 128 ? 128 : (b5 < 1 ? 1 : b5)); }      // (<= 128 atomics per slot)
    const int64_t total = j.kind == 1 ? (int64_t)j.K * j.N : j.kind == 3 ? (int64_t)((j.K + 7) / 8 * 8) * j.N : j.kind == 6 ? (int64_t)(j.K / 8) * j.N : (j.kind == 2 ? (int64_t)j.Kvalid * (((j.T ? j.N : j.K) + 31) / 32 * 32) * 9 : (j.kind == 4 ? (int64_t)j.Kvalid / 8 * (((j.T ? j.N : j.K) + 31) / 32 * 32) * 9 : (int64_t)j.T * j.K * j.N));
    int64_t b = (total + 255) / 256;
    const int64_t cap = j.kind == 1 ? 4096 : 2048;
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

bool push(PnnpPackJob* jobs, int* n, int cap, const PnnpPackJob& j) {
    if (*n >= cap) return false;
    jobs[(*n)++] = j;
    return true;
}

PnnpPackJob gather(const float* src, float* dst, int T, int K, int N, int64_t sk, int64_t sn, int64_t st, int64_t off, int flip,
                   int Kvalid = 1 << 30, int Ndst = 0, int n_off = 0) {
    PnnpPackJob j{};
    j.src = src; j.dst = dst; j.kind = 0; j.T = T; j.K = K; j.N = N; j.sk = sk; j.sn = sn; j.st = st; j.off = off;
    j.flip = flip; j.Kvalid = Kvalid; j.Ndst = Ndst ? Ndst : N; j.n_off = n_off;
    return j;
}

}  // namespace

extern "C" {

// Run `n` pack jobs (HOST array) in ceil(n / 32) launches on `stream`.
int pnnp_pack_jobs_f32(const PnnpPackJob* jobs, int n, void* stream) {
    if (n < 0 || (n && !jobs)) return PNNP_E_INVALID;
    for (int i0 = 0; i0 < n; i0 += MAXJ) {
        JobTable tb{};
        tb.n = n - i0 < MAXJ ? n - i0 : MAXJ;
        int blocks = 0;
        for (int i = 0; i < tb.n; ++i) {
            const PnnpPackJob& j = jobs[i0 + i];
            if (j.kind == 5) { if (!j.src || !j.dst || j.sk < 0) return PNNP_E_INVALID; }
            else if (!j.src || !j.dst || j.K <= 0 || j.N <= 0 || (j.kind == 4 && (!j.amax || j.Kvalid <= 0 || (j.Kvalid & 15))) || (j.kind == 0 && (j.T <= 0 || (j.K & 3))) || (j.kind == 2 && (j.Kvalid <= 0 || (j.Kvalid & 15))) || ((j.kind == 3 || j.kind == 6) && (j.T <= 0 || j.Ndst <= 0 || (j.K & 7))) || (j.kind == 6 && !j.amax)       /* kind 3 enumerates whole 8-row groups */) return PNNP_E_INVALID;
            tb.job[i] = j;
            blocks += job_blocks(j);
            tb.blk_end[i] = blocks;
        }
        hipLaunchKernelGGL(pack_jobs_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), tb);
    }
    return pnnp_launch_status();
}

// Builders: append the jobs of one layer to jobs[0..cap) (HOST array), *n is advanced.  Null destinations are skipped.
// Conv2d weight [Cout][Cin][kh][kw] -> direct forward / backward-data packs (see pnnp_pack_conv_weight_f32).
int pnnp_pack_jobs_add_conv(PnnpPackJob* jobs, int* n, int cap, const float* w, float* fwd, float* dgrad, int Cout, int Cin, int taps,
                            int Cin_pad, int Cout_pad) {
    if (!jobs || !n || !w || Cin_pad < Cin || Cout_pad < Cout) return PNNP_E_INVALID;
    bool ok = true;
    if (fwd) ok = ok && push(jobs, n, cap, gather(w, fwd, taps, Cin_pad, Cout, taps, (int64_t)Cin * taps, 1, 0, 0, Cin));
    if (dgrad) ok = ok && push(jobs, n, cap, gather(w, dgrad, taps, Cout_pad, Cin, (int64_t)Cin * taps, taps, 1, 0, 1, Cout));
    return ok ? PNNP_OK : PNNP_E_WORKSPACE;
}

// ConvTranspose2d weight [Cin][Cout][2][2] (see pnnp_pack_convt_weight_f32).
int pnnp_pack_jobs_add_convt(PnnpPackJob* jobs, int* n, int cap, const float* w, float* fwd, float* dgrad, int Cin, int Cout) {
    if (!jobs || !n || !w) return PNNP_E_INVALID;
    bool ok = true;
    for (int s = 0; s < 4; ++s) {
        if (fwd) ok = ok && push(jobs, n, cap, gather(w, fwd, 1, Cin, Cout, (int64_t)Cout * 4, 4, 0, s, 0, 1 << 30, 4 * Cout, s * Cout));
        if (dgrad) ok = ok && push(jobs, n, cap, gather(w, dgrad + (int64_t)s * Cout * Cin, 1, Cout, Cin, 4, (int64_t)Cout * 4, 0, s, 0));
    }
    return ok ? PNNP_OK : PNNP_E_WORKSPACE;
}

// Winograd filter transforms of a 3x3 Conv2d weight (see pnnp_pack_conv_weight_wino_f32).
int pnnp_pack_jobs_add_wino(PnnpPackJob* jobs, int* n, int cap, const float* w, float* fwd, float* dgrad, int Cout, int Cin) {
    if (!jobs || !n || !w || Cout <= 0 || Cin <= 0) return PNNP_E_INVALID;
    if ((fwd && (Cin % W_KC || Cout % W_BN)) || (dgrad && (Cout % W_KC || Cin % W_BN))) return PNNP_E_UNSUPPORTED;
    bool ok = true;
    for (int d = 0; d < 2; ++d) {
        float* dst = d ? dgrad : fwd;
        if (!dst) continue;
        PnnpPackJob j{};
        j.src = w; j.dst = dst; j.kind = 1; j.T = d; j.K = Cout; j.N = Cin;
        ok = ok && push(jobs, n, cap, j);
    }
    return ok ? PNNP_OK : PNNP_E_WORKSPACE;
}

// bf16x3 packs of a 3x3 Conv2d weight for the pnnp_conv3x3_x3_* kernels (csrc/conv_x3.hip): fwd (K = Cin padded to Cin_pad, a
// multiple of 16; N = Cout) and / or dgrad (K = Cout padded up to a multiple of 16; N = Cin).  N is padded up to a multiple of
// 32 inside the pack (zeros).  Sizes: pnnp_x3_weight_bytes.
int pnnp_pack_jobs_add_x3(PnnpPackJob* jobs, int* n, int cap, const float* w, void* fwd, void* dgrad, int Cout, int Cin, int Cin_pad) {
    if (!jobs || !n || !w || Cout <= 0 || Cin <= 0 || Cin_pad < Cin || (Cin_pad & 15)) return PNNP_E_INVALID;
    bool ok = true;
    for (int d = 0; d < 2; ++d) {
        void* dst = d ? dgrad : fwd;
        if (!dst) continue;
        PnnpPackJob j{};
        j.src = w; j.dst = reinterpret_cast<float*>(dst); j.kind = 2; j.T = d; j.K = Cout; j.N = Cin;
        j.Kvalid = d ? (Cout + 15) / 16 * 16 : Cin_pad;
        // the kernel enumerates N in blocks of 32: K/N of the job are (Cout, Cin); x3_job pads both
        ok = ok && push(jobs, n, cap, j);
    }
    return ok ? PNNP_OK : PNNP_E_WORKSPACE;
}

// fp16x2 packs of a 3x3 Conv2d weight for the pnnp_conv3x3_h2_* kernels (csrc/conv_h2s.hip), same shapes as pnnp_pack_jobs_add_x3; sizes:
// pnnp_h2_weight_bytes.  `amax`: the weight tensor's slot, filled by pnnp_pack_jobs_add_amax in an EARLIER launch of the table (the Python
// PackJobs keeps the amax jobs in a table of their own that runs first).
int pnnp_pack_jobs_add_h2(PnnpPackJob* jobs, int* n, int cap, const float* w, void* fwd, void* dgrad, int Cout, int Cin, int Cin_pad, const unsigned* amax) {
    if (!jobs || !n || !w || !amax || Cout <= 0 || Cin <= 0 || Cin_pad < Cin || (Cin_pad & 15)) return PNNP_E_INVALID;
    bool ok = true;
    for (int d = 0; d < 2; ++d) {
        void* dst = d ? dgrad : fwd;
        if (!dst) continue;
        PnnpPackJob j{};
        j.src = w; j.dst = reinterpret_cast<float*>(dst); j.kind = 4; j.T = d; j.K = Cout; j.N = Cin; j.amax = amax;
        j.Kvalid = d ? (Cout + 15) / 16 * 16 : Cin_pad;
        ok = ok && push(jobs, n, cap, j);
    }
    return ok ? PNNP_OK : PNNP_E_WORKSPACE;
}
// max |x[0 .. count)| -> atomicMax into *slot (the caller zeroes the slot first)
int pnnp_pack_jobs_add_amax(PnnpPackJob* jobs, int* n, int cap, const float* x, int64_t count, unsigned* slot) {
    if (!jobs || !n || !x || !slot || count < 0) return PNNP_E_INVALID;
    PnnpPackJob j{};
    j.src = x; j.dst = reinterpret_cast<float*>(slot); j.kind = 5; j.sk = count; j.K = 1; j.N = 1;
    return push(jobs, n, cap, j) ? PNNP_OK : PNNP_E_WORKSPACE;
}
// bytes of one h2 pack: per (K rounded up to 16) / 16 chunks and (N rounded up to 32) / 32 column blocks 19 taps x 1024 B (9 hi', 9 lo', 1 zero)
int64_t pnnp_h2_weight_bytes(int K, int N) { return (int64_t)((K + 15) / 16) * ((N + 31) / 32) * 19456; }

// max |x[0 .. count)| -> atomicMax into *slot, as a launch of its own: for tensors whose producer has no fused amax (csrc/h2.h)
int pnnp_amax_f32(const float* x, int64_t count, unsigned* slot, void* stream) {
    if (!x || !slot || count < 0) return PNNP_E_INVALID;
    if (count == 0) return PNNP_OK;
    PnnpPackJob j{};
    j.src = x; j.dst = reinterpret_cast<float*>(slot); j.kind = 5; j.sk = count; j.K = 1; j.N = 1;
    return pnnp_pack_jobs_f32(&j, 1, stream);
}

namespace {
PnnpPackJob x3mat(const float* src, void* dst, int K, int N, int64_t sk, int64_t sn, int64_t off, int Ktot, int k_off, int Ntot, int n_off) {
    PnnpPackJob j{};
    j.src = src; j.dst = reinterpret_cast<float*>(dst); j.kind = 3; j.K = K; j.N = N; j.sk = sk; j.sn = sn; j.off = off;
    j.T = Ktot / 16; j.st = k_off; j.Ndst = Ntot; j.n_off = n_off;
    return j;
}
}  // namespace

// x3 packs for the pointwise GEMM kernel (csrc/gemm_x3.hip); all channel counts in multiples of 32.  Sizes: pnnp_x3mat_bytes(K, N).
// ConvTranspose2d weight [Cin][Cout][2][2]: fwd = [K = Cin][N = 4*Cout], column s*Cout + co; dgrad = [K = 4*Cout (row s*Cout + co)][N = Cin]
int pnnp_pack_jobs_add_x3_convt(PnnpPackJob* jobs, int* n, int cap, const float* w, void* fwd, void* dgrad, int Cin, int Cout) {
    if (!jobs || !n || !w || (Cin & 31) || (Cout & 31)) return PNNP_E_INVALID;
    bool ok = true;
    for (int s = 0; s < 4; ++s) {
        if (fwd) ok = ok && push(jobs, n, cap, x3mat(w, fwd, Cin, Cout, (int64_t)Cout * 4, 4, s, Cin, 0, 4 * Cout, s * Cout));
        if (dgrad) ok = ok && push(jobs, n, cap, x3mat(w, dgrad, Cout, Cin, 4, (int64_t)Cout * 4, s, 4 * Cout, s * Cout, Cin, 0));
    }
    return ok ? PNNP_OK : PNNP_E_WORKSPACE;
}
// Conv2d 1x1 weight [Cout][Cin]: fwd = [K = Cin][N = Cout], dgrad = [K = Cout][N = Cin]
int pnnp_pack_jobs_add_x3_1x1(PnnpPackJob* jobs, int* n, int cap, const float* w, void* fwd, void* dgrad, int Cout, int Cin) {
    if (!jobs || !n || !w || (Cin & 31) || (Cout & 31)) return PNNP_E_INVALID;
    bool ok = true;
    if (fwd) ok = ok && push(jobs, n, cap, x3mat(w, fwd, Cin, Cout, 1, Cin, 0, Cin, 0, Cout, 0));
    if (dgrad) ok = ok && push(jobs, n, cap, x3mat(w, dgrad, Cout, Cin, Cin, 1, 0, Cout, 0, Cin, 0));
    return ok ? PNNP_OK : PNNP_E_WORKSPACE;
}
// Conv2d 3x3 stride 2 weight [Cout][Cin][3][3]: fwd = [K = 9*Cin (row t*Cin + ci)][N = Cout]; dgrad = 9 slices [K = Cout][N = Cin], one per
// tap in the parity-class order of pnnp_pack_conv3x3s2_dgrad_f32, the taps of a class stacked along K (class c at the byte offset of its first slice)
int pnnp_pack_jobs_add_x3_s2(PnnpPackJob* jobs, int* n, int cap, const float* w, void* fwd, void* dgrad, int Cout, int Cin) {
    if (!jobs || !n || !w || (Cin & 31) || (Cout & 31)) return PNNP_E_INVALID;
    static const int order[9] = {4, 3, 5, 1, 7, 0, 2, 6, 8};          // S2_TAP_ORDER of csrc/conv_api.hip: classes of 1, 2, 2, 4 taps
    static const int first[9] = {0, 1, 1, 3, 3, 5, 5, 5, 5}, count[9] = {1, 2, 2, 2, 2, 4, 4, 4, 4};
    bool ok = true;
    for (int t = 0; t < 9; ++t) {
        if (fwd) ok = ok && push(jobs, n, cap, x3mat(w, fwd, Cin, Cout, 9, (int64_t)Cin * 9, t, 9 * Cin, t * Cin, Cout, 0));
        // backward-data: one pack per input-pixel parity class, K = (taps of the class) x Cout, the class's taps stacked along K
        if (dgrad) ok = ok && push(jobs, n, cap, x3mat(w, reinterpret_cast<char*>(dgrad) + (int64_t)first[t] * Cout * Cin * 6, Cout, Cin, (int64_t)Cin * 9, 9,
                                                       order[t], count[t] * Cout, (t - first[t]) * Cout, Cin, 0));
    }
    return ok ? PNNP_OK : PNNP_E_WORKSPACE;
}
int64_t pnnp_x3mat_bytes(int K, int N) { return (int64_t)K * N * 6; }
// fp16x2 packs for csrc/gemm_h2s.hip: the same sub-matrix jobs as the x3 builders above in the kind-6 layout (32-channel items: 4 bytes per
// weight), scaled with the weight tensor's amax slot.  Sizes: pnnp_h2mat_bytes(K, N).  Channel counts in multiples of 32.
int pnnp_pack_jobs_add_h2_convt(PnnpPackJob* jobs, int* n, int cap, const float* w, void* fwd, void* dgrad, int Cin, int Cout, const unsigned* amax) {
    if (!jobs || !n || !w || !amax || (Cin & 31) || (Cout & 31)) return PNNP_E_INVALID;
    const int n0 = *n;
    const int rc = pnnp_pack_jobs_add_x3_convt(jobs, n, cap, w, fwd, dgrad, Cin, Cout);
    for (int i = n0; i < *n; ++i) { jobs[i].kind = 6; jobs[i].T /= 2; jobs[i].amax = amax; }       // (T: 16-channel -> 32-channel items)
    return rc;
}
int pnnp_pack_jobs_add_h2_1x1(PnnpPackJob* jobs, int* n, int cap, const float* w, void* fwd, void* dgrad, int Cout, int Cin, const unsigned* amax) {
    if (!jobs || !n || !w || !amax || (Cin & 31) || (Cout & 31)) return PNNP_E_INVALID;
    const int n0 = *n;
    const int rc = pnnp_pack_jobs_add_x3_1x1(jobs, n, cap, w, fwd, dgrad, Cout, Cin);
    for (int i = n0; i < *n; ++i) { jobs[i].kind = 6; jobs[i].T /= 2; jobs[i].amax = amax; }
    return rc;
}
// Conv2d 3x3 stride 2 (layouts of pnnp_pack_jobs_add_x3_s2: fwd = [K = 9 Cin][N = Cout]; dgrad = 9 slices [K = Cout][N = Cin] in parity-class order,
// the taps of a class stacked along K, class c at the byte offset of its first slice: slices of pnnp_h2mat_bytes(Cout, Cin))
int pnnp_pack_jobs_add_h2_s2(PnnpPackJob* jobs, int* n, int cap, const float* w, void* fwd, void* dgrad, int Cout, int Cin, const unsigned* amax) {
    if (!jobs || !n || !w || !amax || (Cin & 31) || (Cout & 31)) return PNNP_E_INVALID;
    static const int order[9] = {4, 3, 5, 1, 7, 0, 2, 6, 8};          // S2_TAP_ORDER of csrc/conv_api.hip: classes of 1, 2, 2, 4 taps
    static const int first[9] = {0, 1, 1, 3, 3, 5, 5, 5, 5}, count[9] = {1, 2, 2, 2, 2, 4, 4, 4, 4};
    bool ok = true;
    const int n0 = *n;
    for (int t = 0; t < 9; ++t) {
        if (fwd) ok = ok && push(jobs, n, cap, x3mat(w, fwd, Cin, Cout, 9, (int64_t)Cin * 9, t, 9 * Cin, t * Cin, Cout, 0));
        if (dgrad) ok = ok && push(jobs, n, cap, x3mat(w, reinterpret_cast<char*>(dgrad) + (int64_t)first[t] * Cout * Cin * 4, Cout, Cin, (int64_t)Cin * 9, 9,
                                                       order[t], count[t] * Cout, (t - first[t]) * Cout, Cin, 0));
    }
    for (int i = n0; i < *n; ++i) { jobs[i].kind = 6; jobs[i].T /= 2; jobs[i].amax = amax; }
    return ok ? PNNP_OK : PNNP_E_WORKSPACE;
}
int64_t pnnp_h2mat_bytes(int K, int N) { return (int64_t)K * N * 4; }


// bytes of one x3 pack: K (reduction channels, rounded up to 16) x N (channels written, rounded up to 32) x 9 taps x 3 pieces x 2 B
int64_t pnnp_x3_weight_bytes(int K, int N) { return (int64_t)((K + 15) / 16 * 16) * ((N + 31) / 32 * 32) * 9 * 6; }

// Backward-data weights of the stride-2 3x3 conv (see pnnp_pack_conv3x3s2_dgrad_f32): 9 slices ordered by input-pixel parity class.
int pnnp_pack_jobs_add_conv3x3s2_dgrad(PnnpPackJob* jobs, int* n, int cap, const float* w, float* dst, int Cout, int Cin) {
    if (!jobs || !n || !w || !dst) return PNNP_E_INVALID;
    static const int order[9] = {4, 3, 5, 1, 7, 0, 2, 6, 8};          // S2_TAP_ORDER of csrc/conv_api.hip
    bool ok = true;
    for (int i = 0; i < 9; ++i)
        ok = ok && push(jobs, n, cap, gather(w, dst + (int64_t)i * Cout * Cin, 1, Cout, Cin, (int64_t)Cin * 9, 9, 0, order[i], 0));
    return ok ? PNNP_OK : PNNP_E_WORKSPACE;
}

}  // extern "C"
