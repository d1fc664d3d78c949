// Model of the INNER LOOP a Winograd F(2x2,3x3) convolution would have on the bf16x3 scheme: not a
// convolution -- the indices are simplified and the results meaningless -- but per 16-channel chunk every wave issues the
// instruction mix the real kernel would need, on random data, with the placement technique of csrc/conv_x3.hip (everything that is
// not an MFMA sits in the gaps between MFMAs, fenced).  What it answers: with the matrix work cut from 36 to 16 multiply-adds per
// output and tap set (2.25x fewer MFMAs), does the REST of the loop -- the input transform B^T d B, the 3-way split of 4x as many
// values as the direct kernel stages, 4x the LDS writes, 8x the weight bytes per MFMA -- still fit under the matrix pipe?
//
// Tile (set by the register file: 16 transformed positions need 16 accumulator sets): 8 x 8 Winograd tiles (16 x 16 output pixels)
// x 64 output channels per 8-wave workgroup = 64 x 16 x 64 floats = 256 KB of the CU's 512 KB of registers.  K chunk = 16 channels.
// LDS cannot hold a whole chunk's operands (V: 64 tiles x 16 xi x 16 ch x 6 B = 96 KB, U: 16 xi x 16 k x 64 n x 6 B = 96 KB), so a
// chunk is four items of 4 xi (one row of the 4 x 4 transform): per item V 24 KB + U 24 KB, double buffered, + the raw fp32 patch.
// Per item and workgroup:   4 xi x (2 x 2 blocks of 32 x 32) x 6 products = 96 MFMAs (12 per wave, v_mfma_f32_32x32x16_bf16)
//   per thread:  6 ds_read_b128 of the raw patch, ~64 VALU (transform of 4 channels x 2 positions + split of 8 values),
//                6 ds_write_b64 of the pieces, 9 operand ds_read_b128 (wave: 2 tile blocks x 3 pieces + 3 weight pieces)
//   per wave:    3 LDS-DMA pieces of 1 KB (U of the next item), raw patch of the next chunk: 3 global loads + 3 LDS stores per chunk
// Direct-convolution equivalent of one chunk: 256 px x 64 ch x 16 ch x 9 taps x 2 = 4.72 MFLOP per CU.
//   hipcc --offload-arch=gfx950 -O3 -o wino_x3_model wino_x3_model.hip && ./wino_x3_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int RAW_BYTES = 324 * 64;                 // 18 x 18 halo pixels x 16 channels fp32
constexpr int V_BYTES = 4 * 3 * 2 * 64 * 16;        // [xi 4][piece 3][octet 2][tile 64][8 bf16]
constexpr int U_BYTES = 4 * 3 * 2 * 64 * 16;        // [xi 4][piece 3][octet 2][n 64][8 bf16]
constexpr int LDS_BYTES = 2 * RAW_BYTES + 2 * V_BYTES + 2 * U_BYTES;     // 139776

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) { unsigned r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void split2(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk_bf16(a0, a1);
    const float r0 = a0 - __uint_as_float(h << 16), r1 = a1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt_pk_bf16(s0, s1);
}

// FLAGS: 1 = transform + split VALU, 2 = raw reads + piece writes (LDS staging traffic), 4 = U by LDS-DMA + raw patch global loads
template <int FLAGS>
__global__ void __launch_bounds__(512, 1) model(const float* __restrict__ act, const unsigned* __restrict__ upack, float* out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* raw = smem; char* vb = smem + 2 * RAW_BYTES; char* ub = vb + 2 * V_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // fill LDS with random data once
    for (int i = tid; i < LDS_BYTES / 16; i += 512) {
        const unsigned h = (i * 2654435761u) ^ (blockIdx.x * 40503u);
        const unsigned w = (0x3f803f80u | (h & 0x007f007fu)) ^ ((h << 7) & 0x80008000u);
        reinterpret_cast<u32x4*>(smem)[i] = u32x4{w, w ^ 0x00110013u, w ^ 0x80010020u, w ^ 0x00408005u};
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc((void*)upack, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)act, 0, 0x7fffffff, 0x00020000);
    f32x16 acc[8];                                  // 4 xi (one per item of the chunk) x 2 tile blocks, for this wave's 32-channel block
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int xi_l = wave >> 1, nb = wave & 1, l31 = lane & 31, half = lane >> 5;
    // staging task of this thread: (tile, channel quad, position pair)
    const int tile = tid & 63, cq = (tid >> 6) & 3, jp = tid >> 8;
    const int ty = tile >> 3, tx = tile & 7;
    f32x4 rg[3];                                    // raw patch of the next chunk in flight
    unsigned uoff = (blockIdx.x & 63) * 98304u;
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cur = g & 1;
            __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0): the LDS-DMA pieces of this item have landed
            __syncthreads();
            const u32x4* vcur = reinterpret_cast<const u32x4*>(vb + cur * V_BYTES);
            const u32x4* ucur = reinterpret_cast<const u32x4*>(ub + cur * U_BYTES);
            char* vnext = vb + (cur ^ 1) * V_BYTES;
            const char* rawc = raw + (c & 1) * RAW_BYTES;
            // operands of this wave: xi_l of the item, tile blocks 0 / 1, channel block nb
            u32x4 a[2][3], b[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[0][p] = vcur[((xi_l * 3 + p) * 2 + half) * 64 + l31];
                a[1][p] = vcur[((xi_l * 3 + p) * 2 + half) * 64 + 32 + l31];
                b[p] = ucur[((xi_l * 3 + p) * 2 + half) * 64 + nb * 32 + l31];
            }
            f32x4 d[2][3], v0, v1; unsigned ph[4], pm[4], pl[4];
            __builtin_amdgcn_sched_barrier(0);
#define MF(I, PA, PB) acc[g * 2 + I] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[I][PA]), __builtin_bit_cast(bf16x8, b[PB]), acc[g * 2 + I], 0, 0, 0)
#define GAP __builtin_amdgcn_sched_barrier(0);
            // 12 MFMAs; the gaps carry the staging of the NEXT item (row (g + 1) & 3 of the transform) and the requests
            MF(0, 0, 2); GAP
            if (FLAGS & 4) {                        // U of the next item: 3 x 1 KB per wave
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, (__attribute__((address_space(3))) void*)(ub + (cur ^ 1) * U_BYTES + (wave * 3 + i) * 1024),
                                                             16, lane * 16u, uoff + (g * 24 + wave * 3 + i) * 1024u, 0, 0);
            }
            GAP MF(0, 2, 0); GAP
            if (FLAGS & 2) {                        // raw rows ty*2 + i, ty*2 + i + 2 (B^T row combination), columns jp .. jp + 2
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    d[0][q] = *reinterpret_cast<const f32x4*>(rawc + (((2 * ty + ((g + 1) & 3)) % 16) * 18 + 2 * tx + jp + q) * 64 + cq * 16);
                }
            }
            GAP MF(0, 1, 1); GAP
            if (FLAGS & 2) {
#pragma unroll
                for (int q = 0; q < 3; ++q) d[1][q] = *reinterpret_cast<const f32x4*>(rawc + (((2 * ty + ((g + 1) & 3)) % 16 + 2) * 18 + 2 * tx + jp + q) * 64 + cq * 16);
            } else {
#pragma unroll
                for (int q = 0; q < 3; ++q) { d[0][q] = __builtin_bit_cast(f32x4, a[0][q]); d[1][q] = __builtin_bit_cast(f32x4, a[1][q]); }
            }
            GAP MF(0, 0, 1); GAP
            if (FLAGS & 1) { d[0][0] -= d[1][0]; d[0][1] -= d[1][1]; }                      // row combination: 3 x 4 VALU
            GAP MF(0, 1, 0); GAP
            if (FLAGS & 1) { d[0][2] -= d[1][2]; v0 = d[0][0] - d[0][2]; }                   // column combinations: 2 x 4 VALU
            GAP MF(0, 0, 0); GAP
            if (FLAGS & 1) { v1 = d[0][1] + d[0][2]; split2(v0[0], v0[1], ph[0], pm[0], pl[0]); }
            GAP MF(1, 0, 2); GAP
            if (FLAGS & 1) split2(v0[2], v0[3], ph[1], pm[1], pl[1]);
            GAP MF(1, 2, 0); GAP
            if (FLAGS & 1) split2(v1[0], v1[1], ph[2], pm[2], pl[2]);
            GAP MF(1, 1, 1); GAP
            if (FLAGS & 1) split2(v1[2], v1[3], ph[3], pm[3], pl[3]);
            else { ph[0] = a[0][0][0]; ph[1] = a[0][0][1]; ph[2] = a[0][1][0]; ph[3] = a[0][1][1]; pm[0] = pm[1] = pm[2] = pm[3] = b[0][0]; pl[0] = pl[1] = pl[2] = pl[3] = b[1][0]; }
            GAP MF(1, 0, 1); GAP
            if (FLAGS & 2) {                        // 2 positions x 3 pieces x 4 channels (8 bytes) each
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    char* dst = vnext + ((((2 * jp + j) * 3) * 2 + (cq >> 1)) * 64 + tile) * 16 + (cq & 1) * 8;
                    *reinterpret_cast<u32x2*>(dst) = u32x2{ph[2 * j], ph[2 * j + 1]};
                    *reinterpret_cast<u32x2*>(dst + 2 * 64 * 16) = u32x2{pm[2 * j], pm[2 * j + 1]};
                    if (j == 0) *reinterpret_cast<u32x2*>(dst + 4 * 64 * 16) = u32x2{pl[0], pl[1]};
                }
            }
            GAP MF(1, 1, 0); GAP
            if (FLAGS & 2) {
                char* dst = vnext + ((((2 * jp + 1) * 3) * 2 + (cq >> 1)) * 64 + tile) * 16 + (cq & 1) * 8;
                *reinterpret_cast<u32x2*>(dst + 4 * 64 * 16) = u32x2{pl[2], pl[3]};
            }
            if ((FLAGS & 4) && g == 0) {            // raw patch of the next chunk: 1296 x 16 B over 512 threads
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int s = tid + 512 * i;
                    rg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa, s < 1296 ? (unsigned)s * 16u : 0x80000000u, ((c * 977 + blockIdx.x * 131) & 4095) * 20736, 0));
                }
            }
            GAP MF(1, 0, 0); GAP
            if ((FLAGS & 4) && g == 3) {
                __builtin_amdgcn_s_waitcnt(0x0f70 | 3);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int s = tid + 512 * i;
                    if (s < 1296) *reinterpret_cast<f32x4*>(raw + ((c + 1) & 1) * RAW_BYTES + s * 16) = rg[i];
                }
            }
#undef MF
#undef GAP
        }
        uoff = (uoff + 98304u) & 0x01ffffffu;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + tid] = s;
}

template <int FLAGS>
void run(const char* name, const float* act, const unsigned* upack, float* out, int chunks) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(model<FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0, best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((model<FLAGS>), dim3(256), dim3(512), LDS_BYTES, 0, act, upack, out, chunks);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double alg = 256.0 * chunks * 256.0 * 64 * 16 * 9 * 2;          // direct-convolution FLOP the chunks stand for
    const double exe = 256.0 * chunks * 384.0 * 32768;                    // bf16 FLOP the matrix pipe executed (384 MFMAs per chunk and CU)
    printf("%-58s %8.3f ms  %6.0f cycles/chunk at 2.4 GHz  conv-equivalent %6.1f TFLOP/s  matrix pipe %5.1f %% of 2516.6\n", name, best,
           best * 1e-3 * 2.4e9 / chunks, alg / best / 1e9, exe / best / 1e9 / 25.166);
}

int main() {
    const int chunks = 4000;
    float* act; unsigned* upack; float* out;
    const size_t abytes = 4096ull * 20736 + 65536, ubytes = 64ull << 20;
    hipMalloc(&act, abytes); hipMalloc(&upack, ubytes); hipMalloc(&out, 256 * 512 * 4);
    unsigned* h = (unsigned*)malloc(ubytes);
    for (size_t i = 0; i < ubytes / 4; ++i) { const unsigned x = (unsigned)i * 2654435761u; h[i] = (0x3f803f80u | (x & 0x007f007fu)) ^ ((x << 5) & 0x80008000u); }
    hipMemcpy(upack, h, ubytes, hipMemcpyHostToDevice);
    float* ha = (float*)h;
    for (size_t i = 0; i < abytes / 4; ++i) ha[i] = (float)((i * 2654435761u) >> 8) * (1.0f / 16777216.0f) - 0.5f;
    hipMemcpy(act, ha, abytes, hipMemcpyHostToDevice);
    run<0>("MFMAs + operand reads only (2.25x fewer MFMAs than direct)", act, upack, out, chunks);
    run<4>(" + U by LDS-DMA (24 KB / item) and the raw patch loads", act, upack, out, chunks);
    run<6>(" + raw reads and piece writes (LDS staging traffic)", act, upack, out, chunks);
    run<7>(" + input transform and 3-way split (the full loop)", act, upack, out, chunks);
    run<3>("full loop without the global / DMA traffic", act, upack, out, chunks);
    run<7>(" + input transform and 3-way split (the full loop), again", act, upack, out, chunks);
    printf("reference points: igemm_x3_kernel on conv4_2 / conv6_1 forward (whole layer, 16 x 64 x 64): 243 / 252 conv-equivalent TFLOP/s;\n"
           "a Winograd kernel would also pay the output transform A^T M A + epilogue per tile and the filter transform U = G g G^T per step.\n");
    return 0;
}
