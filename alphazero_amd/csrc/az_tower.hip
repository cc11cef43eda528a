// Dual-head residual tower, inference only (AlphaZero/network/model.py:39-88 with
// batch-norm folded into the convolutions: network/util.py:18-21, center only).
//
// Activations are NHWC with a one-pixel zero halo per board: [B][(H+2)*(W+2)][C].
// A 3x3 SAME convolution is then an implicit GEMM  D[m][co] = sum_{tap,ci} X[pix(m)+off(tap)][ci] * Wt[co][tap][ci]
// with no boundary tests in the inner loop; the halo is never written.
// bf16 path: MFMA 32x32x16 (fp32 accumulate), fused bias + residual + ReLU epilogue.
// fp32 path: same data flow on the vector ALU, used for the 1e-5 parity mode.
#include "az_tower.h"

#include <hip/hip_bf16.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short bf16_t;

#define CIN0 32  // input planes (17) padded to one K chunk

struct Tower {
    int H, W, HW, PP, O, NB, C, dtype, maxB;
    size_t blob_floats;
    bool ready;
    std::vector<void *> allocs;
    void *w_stem, *w_conv;          // [C][9][CIN0], [2*NB][C][9][C] in the compute dtype
    float *b_stem, *b_conv;         // [C], [2*NB][C]
    float *w_head, *b_head;         // [3][C], [3]
    float *w_pfc, *b_pfc;           // [2*HW][O], [O]
    float *w_v1, *b_v1, *w_v2, *b_v2;  // [HW][256], [256], [256], [1]
    void *X0, *X, *T, *Y;           // activations
    float *hp, *hv;                 // [maxB][2*HW], [maxB][HW]
    // per-launch timing of the residual-tower convolutions (HIP events on the launch stream)
    bool profile;
    std::vector<hipEvent_t> evs;    // 2 per conv launch of one forward
    int evs_used;
    double conv_ms;
    unsigned long long conv_launches;
};

__device__ __forceinline__ bf16_t f2bf(float f)
{
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((unsigned int)b) << 16); }

static inline bf16_t h_f2bf(float f)
{
    unsigned int u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// ---------------------------------------------------------------------------
// int8 planes (B,17,H,W) -> NHWC with halo, CIN0 channels (17 real + zeros)
// ---------------------------------------------------------------------------
template <typename T>
__global__ void k_planes_to_nhwc(const int8_t *planes, T *X0, int H, int W)
{
    const int b = blockIdx.x, HW = H * W, PP = (H + 2) * (W + 2);
    const int8_t *src = planes + (size_t)b * 17 * HW;
    for (int idx = threadIdx.x; idx < HW * CIN0; idx += blockDim.x) {
        int pix = idx / CIN0, c = idx - pix * CIN0;
        int y = pix / W, x = pix - y * W;
        float v = c < 17 ? (float)src[c * HW + pix] : 0.0f;
        size_t o = ((size_t)b * PP + (y + 1) * (W + 2) + (x + 1)) * CIN0 + c;
        if constexpr (sizeof(T) == 2) X0[o] = f2bf(v); else X0[o] = v;
    }
}

// ---------------------------------------------------------------------------
// bf16 implicit-GEMM 3x3 convolution on MFMA.
// Workgroup tile 128 pixels x 128 output channels, 4 waves as 2x2, each wave a
// 64x64 tile = 2x2 MFMA 32x32 accumulators; K step = one tap x KC channels.
// ---------------------------------------------------------------------------
template <int KC, bool RESID>
__global__ __launch_bounds__(256) void k_conv3x3_bf16(const bf16_t *__restrict__ X, const bf16_t *__restrict__ Wt,
                                                      const float *__restrict__ bias, const bf16_t *__restrict__ R,
                                                      bf16_t *__restrict__ Y, int M, int HW, int Wd, int PP, int CIN, int COUT)
{
    constexpr int LDK = KC + 8;          // +16 B per row: conflict-free 16-byte fragment reads
    constexpr int CPR = KC / 8;          // 16-byte chunks per row
    constexpr int RPP = 256 / CPR;       // rows loaded per pass
    constexpr int NP = 128 / RPP;        // passes
    __shared__ __attribute__((aligned(16))) bf16_t As[128 * LDK];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[128 * LDK];
    __shared__ int rowoff[128];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
    const int W2 = Wd + 2;

    if (tid < 128) {
        int m = min(m0 + tid, M - 1);
        int b = m / HW, pix = m - b * HW, y = pix / Wd, x = pix - y * Wd;
        rowoff[tid] = b * PP + (y + 1) * W2 + (x + 1);
    }
    __syncthreads();
    const int lrow = tid / CPR, lch = tid - lrow * CPR;
    size_t abase[NP], wbase[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        abase[p] = (size_t)rowoff[lrow + p * RPP] * CIN + lch * 8;
        wbase[p] = (size_t)(n0 + lrow + p * RPP) * 9 * CIN + lch * 8;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    for (int tap = 0; tap < 9; tap++) {
        const long toff = (long)((tap / 3 - 1) * W2 + (tap % 3 - 1)) * CIN;
        for (int c0 = 0; c0 < CIN; c0 += KC) {
            uint4 ra[NP], rb[NP];
#pragma unroll
            for (int p = 0; p < NP; p++) {
                ra[p] = *reinterpret_cast<const uint4 *>(X + (long)abase[p] + toff + c0);
                rb[p] = *reinterpret_cast<const uint4 *>(Wt + wbase[p] + (size_t)tap * CIN + c0);
            }
            __syncthreads();
#pragma unroll
            for (int p = 0; p < NP; p++) {
                *reinterpret_cast<uint4 *>(&As[(lrow + p * RPP) * LDK + lch * 8]) = ra[p];
                *reinterpret_cast<uint4 *>(&Bs[(lrow + p * RPP) * LDK + lch * 8]) = rb[p];
            }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < KC / 16; ks++) {
                bf16x8 af[2], bf[2];
                const int kk = ks * 16 + 8 * (lane >> 5);
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    af[i] = *reinterpret_cast<const bf16x8 *>(&As[(wr * 64 + i * 32 + (lane & 31)) * LDK + kk]);
                    bf[i] = *reinterpret_cast<const bf16x8 *>(&Bs[(wc * 64 + i * 32 + (lane & 31)) * LDK + kk]);
                }
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    // epilogue: + bias (+ residual), ReLU, bf16, NHWC store
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int n = n0 + wc * 64 + j * 32 + (lane & 31);
        const float bn = bias[n];
#pragma unroll
        for (int i = 0; i < 2; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m0 + row < M) {
                    size_t o = (size_t)rowoff[row] * COUT + n;
                    float v = acc[i][j][r] + bn;
                    if (RESID) v += bf2f(R[o]);
                    Y[o] = f2bf(fmaxf(v, 0.0f));
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// bf16 implicit-GEMM 3x3 convolution, 256 -> 256 channels: the tower's hot kernel.
//
// Workgroup = 512 threads (8 waves) computes a 256-pixel x 256-channel output tile.
// GEMM orientation: MFMA A operand = weights (rows = output channels), B operand =
// activations (columns = pixels), so every lane ends up with 4 consecutive channels
// of one pixel per accumulator.  K step = one tap x 64 input channels (36 steps).
// Both operand tiles live in LDS as [256 rows][64 bf16] (128 B rows, no padding) in
// two buffers; they are filled by global_load_lds_dwordx4 (16 B per lane, 8 rows per
// wave instruction).  The 16-byte slot s of row r holds k-chunk s ^ ((r >> 1) & 7):
// the XOR is applied to the per-lane GLOBAL address at fill time and to the LDS
// address at read time, which makes every ds_read_b128 lane group hit 64 distinct
// banks.  Wave (wr, wc) owns pixels wr*128.. and channels wc*64..: 4 x 8 MFMA
// 16x16x32 accumulators.
// Epilogue: + bias -> bf16 -> per-wave LDS transpose ([pixel][channel], 144 B rows)
// -> 16-byte row reads, + residual, ReLU, 16-byte coalesced NHWC stores.
// ---------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define V2_LDS_BYTES (147456 + 1024)

__device__ __forceinline__ void glds16(const bf16_t *g, unsigned char *l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

template <bool RESID>
__global__ __launch_bounds__(512) void k_conv3x3_c256_bf16(const bf16_t *__restrict__ X, const bf16_t *__restrict__ Wt,
                                                           const float *__restrict__ bias, const bf16_t *__restrict__ R,
                                                           bf16_t *__restrict__ Y, int M, int HW, int Wd, int PP)
{
    constexpr int C = 256;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[V2_LDS_BYTES];
    int *rowoff = reinterpret_cast<int *>(smem + 147456);  // padded-pixel index of each tile row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int m0 = blockIdx.x * 256, W2 = Wd + 2;
    if (tid < 256) {
        int m = min(m0 + tid, M - 1);
        int b = m / HW, pix = m - b * HW, y = pix / Wd, x = pix - y * Wd;
        rowoff[tid] = b * PP + (y + 1) * W2 + (x + 1);
    }
    __syncthreads();
    // fill mapping: wave w fills rows w*32 + q*8 + (lane>>3), slot lane&7, of both tiles
    const int frow = lane >> 3, fslot = lane & 7;
    int xoff[4], woff[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int r = wave * 32 + q * 8 + frow;
        const int kc = fslot ^ ((r >> 1) & 7);
        xoff[q] = rowoff[r] * C + kc * 8;
        woff[q] = r * 9 * C + kc * 8;
    }
    unsigned char *const fill_base = smem + (wave * 32) * 128;  // + q*1024 (+ buffer, + tile)
    // read mapping (same for both tiles): row (lane&15) of a 16-row group, k-chunk ks2*4 + (lane>>4)
    const int rd0 = (lane & 15) * 128 + ((((lane >> 4)) ^ ((lane & 15) >> 1)) << 4);
    const int rd1 = (lane & 15) * 128 + ((((lane >> 4) + 4) ^ ((lane & 15) >> 1)) << 4);
    const unsigned char *const wrd = smem + (wc * 64) * 128;
    const unsigned char *const xrd = smem + 32768 + (wr * 128) * 128;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto issue = [&](int step, int buf) {
        const int tap = step >> 2, c0 = (step & 3) * 64;
        const int ty = tap / 3, tx = tap - ty * 3;
        const int toff = ((ty - 1) * W2 + (tx - 1)) * C + c0;
        const int wofs = tap * C + c0;
        unsigned char *dst = fill_base + buf * 65536;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            glds16(Wt + woff[q] + wofs, dst + q * 1024);
            glds16(X + xoff[q] + toff, dst + 32768 + q * 1024);
        }
    };

    issue(0, 0);
    for (int step = 0; step < 36; step++) {
        const int buf = step & 1;
        __syncthreads();  // drains this wave's LDS-DMA (vmcnt 0) and fences everybody's reads of the other buffer
        if (step + 1 < 36) issue(step + 1, buf ^ 1);
        const unsigned char *wb = wrd + buf * 65536;
        const unsigned char *xb = xrd + buf * 65536;
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ks2++) {
            const int rd = ks2 ? rd1 : rd0;
            bf16x8 wf[4], xf[8];
#pragma unroll
            for (int i = 0; i < 4; i++) wf[i] = *reinterpret_cast<const bf16x8 *>(wb + i * 2048 + rd);
#pragma unroll
            for (int j = 0; j < 8; j++) xf[j] = *reinterpret_cast<const bf16x8 *>(xb + j * 2048 + rd);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 8; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    // ---- epilogue: per-wave transpose through LDS
    unsigned char *reg = smem + wave * 18432;  // [128 pixels][144 B]
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float4 b4 = *reinterpret_cast<const float4 *>(bias + wc * 64 + i * 16 + (lane >> 4) * 4);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const f32x4 a = acc[i][j];
            uint2 pk;
            pk.x = (unsigned)f2bf(a[0] + b4.x) | ((unsigned)f2bf(a[1] + b4.y) << 16);
            pk.y = (unsigned)f2bf(a[2] + b4.z) | ((unsigned)f2bf(a[3] + b4.w) << 16);
            *reinterpret_cast<uint2 *>(reg + (j * 16 + (lane & 15)) * 144 + (i * 16 + (lane >> 4) * 4) * 2) = pk;
        }
    }
    __syncthreads();
    const int erow = lane >> 3, echunk = lane & 7;
#pragma unroll 4
    for (int it = 0; it < 16; it++) {
        const int row = it * 8 + erow;
        const int trow = wr * 128 + row;
        if (m0 + trow < M) {
            uint4 v = *reinterpret_cast<const uint4 *>(reg + row * 144 + echunk * 16);
            const size_t o = (size_t)rowoff[trow] * C + wc * 64 + echunk * 8;
            unsigned vv[4] = {v.x, v.y, v.z, v.w};
            unsigned rr[4] = {0, 0, 0, 0};
            if (RESID) {
                uint4 r4 = *reinterpret_cast<const uint4 *>(R + o);
                rr[0] = r4.x; rr[1] = r4.y; rr[2] = r4.z; rr[3] = r4.w;
            }
            unsigned out[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float lo = bf2f((bf16_t)(vv[k] & 0xffff)), hi = bf2f((bf16_t)(vv[k] >> 16));
                if (RESID) { lo += bf2f((bf16_t)(rr[k] & 0xffff)); hi += bf2f((bf16_t)(rr[k] >> 16)); }
                out[k] = (unsigned)f2bf(fmaxf(lo, 0.f)) | ((unsigned)f2bf(fmaxf(hi, 0.f)) << 16);
            }
            *reinterpret_cast<uint4 *>(Y + o) = make_uint4(out[0], out[1], out[2], out[3]);
        }
    }
}

// ---------------------------------------------------------------------------
// fp32 implicit-GEMM 3x3 convolution (vector ALU): 64 pixels x 64 channels per
// workgroup, 4x4 outputs per thread, K step = one tap x 16 channels.
// ---------------------------------------------------------------------------
template <bool RESID>
__global__ __launch_bounds__(256) void k_conv3x3_f32(const float *__restrict__ X, const float *__restrict__ Wt,
                                                     const float *__restrict__ bias, const float *__restrict__ R,
                                                     float *__restrict__ Y, int M, int HW, int Wd, int PP, int CIN, int COUT)
{
    __shared__ float As[16][64 + 4];
    __shared__ float Bs[16][64 + 4];
    __shared__ int rowoff[64];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64, W2 = Wd + 2;
    if (tid < 64) {
        int m = min(m0 + tid, M - 1);
        int b = m / HW, pix = m - b * HW, y = pix / Wd, x = pix - y * Wd;
        rowoff[tid] = b * PP + (y + 1) * W2 + (x + 1);
    }
    __syncthreads();
    // loader: 64 rows x 16 k = 1024 floats, 4 per thread (one float4 along k)
    const int lrow = tid >> 2, lk = (tid & 3) * 4;
    const size_t abase = (size_t)rowoff[lrow] * CIN + lk;
    const size_t wbase = (size_t)(n0 + lrow) * 9 * CIN + lk;
    float acc[4][4] = {};
    for (int tap = 0; tap < 9; tap++) {
        const long toff = (long)((tap / 3 - 1) * W2 + (tap % 3 - 1)) * CIN;
        for (int c0 = 0; c0 < CIN; c0 += 16) {
            float4 a = *reinterpret_cast<const float4 *>(X + (long)abase + toff + c0);
            float4 w = *reinterpret_cast<const float4 *>(Wt + wbase + (size_t)tap * CIN + c0);
            __syncthreads();
            As[lk + 0][lrow] = a.x; As[lk + 1][lrow] = a.y; As[lk + 2][lrow] = a.z; As[lk + 3][lrow] = a.w;
            Bs[lk + 0][lrow] = w.x; Bs[lk + 1][lrow] = w.y; Bs[lk + 2][lrow] = w.z; Bs[lk + 3][lrow] = w.w;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; k++) {
                float av[4], bv[4];
#pragma unroll
                for (int i = 0; i < 4; i++) { av[i] = As[k][ty * 4 + i]; bv[i] = Bs[k][tx * 4 + i]; }
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = ty * 4 + i;
        if (m0 + row >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = n0 + tx * 4 + j;
            size_t o = (size_t)rowoff[row] * COUT + n;
            float v = acc[i][j] + bias[n];
            if (RESID) v += R[o];
            Y[o] = fmaxf(v, 0.0f);
        }
    }
}

// ---------------------------------------------------------------------------
// heads.  1x1 convolutions 256->2 (policy) and 256->1 (value) with folded BN + ReLU:
// one wave per pixel, lanes across channels.  Output in NCHW-flatten order
// (model.py:71-72 reshapes an NCHW tensor), i.e. hp[b][c*HW + pix].
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_head_conv(const T *__restrict__ X, const float *__restrict__ wh, const float *__restrict__ bh,
                                                   float *__restrict__ hp, float *__restrict__ hv, int HW, int Wd, int PP, int C)
{
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int W2 = Wd + 2;
    const float b0 = bh[0], b1 = bh[1], b2 = bh[2];
    for (int pix = wave; pix < HW; pix += 4) {
        int y = pix / Wd, x = pix - y * Wd;
        const T *row = X + ((size_t)b * PP + (y + 1) * W2 + (x + 1)) * C;
        float s0 = 0, s1 = 0, s2 = 0;
        for (int c = lane; c < C; c += 64) {
            float v;
            if constexpr (sizeof(T) == 2) v = bf2f(row[c]); else v = row[c];
            s0 = fmaf(v, wh[c], s0); s1 = fmaf(v, wh[C + c], s1); s2 = fmaf(v, wh[2 * C + c], s2);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64);
        }
        if (lane == 0) {
            hp[(size_t)b * 2 * HW + pix] = fmaxf(s0 + b0, 0.0f);
            hp[(size_t)b * 2 * HW + HW + pix] = fmaxf(s1 + b1, 0.0f);
            hv[(size_t)b * HW + pix] = fmaxf(s2 + b2, 0.0f);
        }
    }
}

// policy FC + softmax, value FC256 + ReLU + FC1 + tanh; one workgroup per board
__global__ __launch_bounds__(256) void k_head_fc(const float *__restrict__ hp, const float *__restrict__ hv,
                                                 const float *__restrict__ wp, const float *__restrict__ bp,
                                                 const float *__restrict__ w1, const float *__restrict__ b1,
                                                 const float *__restrict__ w2, const float *__restrict__ b2,
                                                 float *__restrict__ priors, float *__restrict__ values, int HW, int O)
{
    __shared__ float sp[2 * 361];
    __shared__ float sv[361];
    __shared__ float red[256];
    __shared__ float logit[384];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < 2 * HW; i += 256) sp[i] = hp[(size_t)b * 2 * HW + i];
    for (int i = tid; i < HW; i += 256) sv[i] = hv[(size_t)b * HW + i];
    __syncthreads();
    float lmax = -INFINITY;
    for (int o = tid; o < O; o += 256) {
        float s = 0.0f;
        for (int j = 0; j < 2 * HW; j++) s = fmaf(sp[j], wp[(size_t)j * O + o], s);
        s += bp[o];
        logit[o] = s;
        lmax = fmaxf(lmax, s);
    }
    red[tid] = lmax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]); __syncthreads(); }
    const float mx = red[0];
    __syncthreads();
    float lsum = 0.0f;
    for (int o = tid; o < O; o += 256) { float e = expf(logit[o] - mx); logit[o] = e; lsum += e; }
    red[tid] = lsum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const float tot = red[0];
    __syncthreads();
    for (int o = tid; o < O; o += 256) priors[(size_t)b * O + o] = logit[o] / tot;
    // value head
    float h = 0.0f;
    for (int j = 0; j < HW; j++) h = fmaf(sv[j], w1[(size_t)j * 256 + tid], h);
    h = fmaxf(h + b1[tid], 0.0f);
    red[tid] = h * w2[tid];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    if (tid == 0) values[b] = tanhf(red[0] + b2[0]);
}

// ---------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------

template <typename T>
static T *talloc(Tower *t, size_t n)
{
    void *p = nullptr;
    if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
    hipMemset(p, 0, (n ? n : 1) * sizeof(T));
    t->allocs.push_back(p);
    return reinterpret_cast<T *>(p);
}

size_t tower_blob_floats(const Tower *t) { return t->blob_floats; }
bool tower_ready(const Tower *t) { return t->ready; }

double tower_flops_per_eval(const Tower *t)
{
    // 2*H*W*[9*17*C + n_conv*9*C^2 + C*2 + C*1] + 2*[2HW*O + HW*256 + 256]
    const double HW = t->HW, C = t->C;
    return 2.0 * HW * (9.0 * 17.0 * C + (2.0 * t->NB) * 9.0 * C * C + C * 3.0) + 2.0 * (2.0 * HW * t->O + HW * 256.0 + 256.0);
}

double tower_conv_flops_per_launch(const Tower *t, int B) { return 2.0 * (double)B * t->HW * 9.0 * t->C * t->C; }

Tower *tower_create(int H, int W, int O, int num_blocks, int channels, int dtype, int max_batch, std::string &err)
{
    if (channels % 128 != 0) { err = "channels must be a multiple of 128"; return nullptr; }
    if (H * W > 361) { err = "board too large"; return nullptr; }
    Tower *t = new Tower();
    t->H = H; t->W = W; t->HW = H * W; t->PP = (H + 2) * (W + 2); t->O = O; t->NB = num_blocks; t->C = channels;
    t->dtype = dtype; t->maxB = max_batch; t->ready = false;
    t->profile = false; t->evs_used = 0; t->conv_ms = 0; t->conv_launches = 0;
    const size_t C = channels, HW = t->HW;
    t->blob_floats = C * 9 * 17 + C + (size_t)num_blocks * 2 * (C * 9 * C + C) + 3 * C + 3 + 2 * HW * O + O + HW * 256 + 256 + 256 + 1;
    const size_t es = dtype == 1 ? 2 : 4;
    bool ok = true;
    ok = ok && (t->w_stem = talloc<char>(t, C * 9 * CIN0 * es));
    ok = ok && (t->w_conv = talloc<char>(t, (size_t)num_blocks * 2 * C * 9 * C * es));
    ok = ok && (t->b_stem = talloc<float>(t, C)) && (t->b_conv = talloc<float>(t, (size_t)num_blocks * 2 * C));
    ok = ok && (t->w_head = talloc<float>(t, 3 * C)) && (t->b_head = talloc<float>(t, 3));
    ok = ok && (t->w_pfc = talloc<float>(t, 2 * HW * O)) && (t->b_pfc = talloc<float>(t, O));
    ok = ok && (t->w_v1 = talloc<float>(t, HW * 256)) && (t->b_v1 = talloc<float>(t, 256));
    ok = ok && (t->w_v2 = talloc<float>(t, 256)) && (t->b_v2 = talloc<float>(t, 1));
    const size_t act = (size_t)max_batch * t->PP;
    // one extra board of slack on both sides is not needed: tap offsets stay inside [0, PP) of a board's own halo
    ok = ok && (t->X0 = talloc<char>(t, act * CIN0 * es));
    ok = ok && (t->X = talloc<char>(t, act * C * es)) && (t->T = talloc<char>(t, act * C * es)) && (t->Y = talloc<char>(t, act * C * es));
    ok = ok && (t->hp = talloc<float>(t, (size_t)max_batch * 2 * HW)) && (t->hv = talloc<float>(t, (size_t)max_batch * HW));
    if (!ok) { err = "device allocation failed"; tower_destroy(t); return nullptr; }
    return t;
}

void tower_destroy(Tower *t)
{
    if (!t) return;
    for (void *p : t->allocs) hipFree(p);
    for (hipEvent_t e : t->evs) hipEventDestroy(e);
    delete t;
}

static int upload_conv(void *dst, const float *src, size_t n, int dtype, hipStream_t st)
{
    if (dtype == 1) {
        std::vector<bf16_t> tmp(n);
        for (size_t i = 0; i < n; i++) tmp[i] = h_f2bf(src[i]);
        if (hipMemcpyAsync(dst, tmp.data(), n * 2, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
        return hipStreamSynchronize(st) == hipSuccess ? 0 : -1;
    }
    if (hipMemcpyAsync(dst, src, n * 4, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    return hipStreamSynchronize(st) == hipSuccess ? 0 : -1;
}

int tower_load(Tower *t, const float *blob, size_t n, hipStream_t st, std::string &err)
{
    if (n != t->blob_floats) {
        char buf[128];
        snprintf(buf, sizeof buf, "weight blob has %zu floats, expected %zu", n, t->blob_floats);
        err = buf;
        return -1;
    }
    const size_t C = t->C, HW = t->HW, O = t->O;
    const float *p = blob;
    int rc = 0;
    {   // stem [C][9][17] -> [C][9][CIN0] zero padded
        std::vector<float> w(C * 9 * CIN0, 0.0f);
        for (size_t co = 0; co < C; co++)
            for (int tap = 0; tap < 9; tap++)
                for (int ci = 0; ci < 17; ci++) w[(co * 9 + tap) * CIN0 + ci] = p[(co * 9 + tap) * 17 + ci];
        rc |= upload_conv(t->w_stem, w.data(), w.size(), t->dtype, st);
        p += C * 9 * 17;
        rc |= upload_conv(t->b_stem, p, C, 0, st);
        p += C;
    }
    for (int l = 0; l < 2 * t->NB; l++) {
        rc |= upload_conv((char *)t->w_conv + (size_t)l * C * 9 * C * (t->dtype == 1 ? 2 : 4), p, C * 9 * C, t->dtype, st);
        p += C * 9 * C;
        rc |= upload_conv(t->b_conv + (size_t)l * C, p, C, 0, st);
        p += C;
    }
    rc |= upload_conv(t->w_head, p, 3 * C, 0, st); p += 3 * C;
    rc |= upload_conv(t->b_head, p, 3, 0, st); p += 3;
    rc |= upload_conv(t->w_pfc, p, 2 * HW * O, 0, st); p += 2 * HW * O;
    rc |= upload_conv(t->b_pfc, p, O, 0, st); p += O;
    rc |= upload_conv(t->w_v1, p, HW * 256, 0, st); p += HW * 256;
    rc |= upload_conv(t->b_v1, p, 256, 0, st); p += 256;
    rc |= upload_conv(t->w_v2, p, 256, 0, st); p += 256;
    rc |= upload_conv(t->b_v2, p, 1, 0, st); p += 1;
    if (rc) { err = "weight upload failed"; return -1; }
    t->ready = true;
    return 0;
}

template <typename T>
static void launch_conv(const Tower *t, const T *X, const T *Wt, const float *bias, const T *R, T *Y, int B, int cin, hipStream_t st);

template <>
void launch_conv<bf16_t>(const Tower *t, const bf16_t *X, const bf16_t *Wt, const float *bias, const bf16_t *R, bf16_t *Y, int B,
                         int cin, hipStream_t st)
{
    const int M = B * t->HW;
    dim3 grid((M + 127) / 128, t->C / 128), block(256);
    if (cin == 256 && t->C == 256) {
        dim3 g2((M + 255) / 256), b2(512);
        if (R) hipLaunchKernelGGL((k_conv3x3_c256_bf16<true>), g2, b2, 0, st, X, Wt, bias, R, Y, M, t->HW, t->W, t->PP);
        else hipLaunchKernelGGL((k_conv3x3_c256_bf16<false>), g2, b2, 0, st, X, Wt, bias, R, Y, M, t->HW, t->W, t->PP);
        return;
    }
    if (cin == CIN0)
        hipLaunchKernelGGL((k_conv3x3_bf16<32, false>), grid, block, 0, st, X, Wt, bias, R, Y, M, t->HW, t->W, t->PP, cin, t->C);
    else if (R)
        hipLaunchKernelGGL((k_conv3x3_bf16<64, true>), grid, block, 0, st, X, Wt, bias, R, Y, M, t->HW, t->W, t->PP, cin, t->C);
    else
        hipLaunchKernelGGL((k_conv3x3_bf16<64, false>), grid, block, 0, st, X, Wt, bias, R, Y, M, t->HW, t->W, t->PP, cin, t->C);
}

template <>
void launch_conv<float>(const Tower *t, const float *X, const float *Wt, const float *bias, const float *R, float *Y, int B, int cin,
                        hipStream_t st)
{
    const int M = B * t->HW;
    dim3 grid((M + 63) / 64, t->C / 64), block(256);
    if (R)
        hipLaunchKernelGGL((k_conv3x3_f32<true>), grid, block, 0, st, X, Wt, bias, R, Y, M, t->HW, t->W, t->PP, cin, t->C);
    else
        hipLaunchKernelGGL((k_conv3x3_f32<false>), grid, block, 0, st, X, Wt, bias, R, Y, M, t->HW, t->W, t->PP, cin, t->C);
}

template <typename T>
static int forward_t(Tower *t, const int8_t *planes, int B, float *priors, float *values, hipStream_t st)
{
    const size_t C = t->C;
    T *X0 = (T *)t->X0, *X = (T *)t->X, *Tm = (T *)t->T, *Y = (T *)t->Y;
    hipLaunchKernelGGL((k_planes_to_nhwc<T>), dim3(B), dim3(256), 0, st, planes, X0, t->H, t->W);
    launch_conv<T>(t, X0, (const T *)t->w_stem, t->b_stem, (const T *)nullptr, X, B, CIN0, st);
    for (int l = 0; l < t->NB; l++) {
        const T *w1 = (const T *)t->w_conv + (size_t)(2 * l) * C * 9 * C;
        const T *w2 = (const T *)t->w_conv + (size_t)(2 * l + 1) * C * 9 * C;
        if (t->profile) hipEventRecord(t->evs[t->evs_used++], st);
        launch_conv<T>(t, X, w1, t->b_conv + (size_t)(2 * l) * C, (const T *)nullptr, Tm, B, (int)C, st);
        if (t->profile) { hipEventRecord(t->evs[t->evs_used++], st); hipEventRecord(t->evs[t->evs_used++], st); }
        launch_conv<T>(t, Tm, w2, t->b_conv + (size_t)(2 * l + 1) * C, X, Y, B, (int)C, st);
        if (t->profile) hipEventRecord(t->evs[t->evs_used++], st);
        T *tmp = X; X = Y; Y = tmp;
    }
    hipLaunchKernelGGL((k_head_conv<T>), dim3(B), dim3(256), 0, st, (const T *)X, t->w_head, t->b_head, t->hp, t->hv, t->HW, t->W,
                       t->PP, (int)C);
    hipLaunchKernelGGL(k_head_fc, dim3(B), dim3(256), 0, st, t->hp, t->hv, t->w_pfc, t->b_pfc, t->w_v1, t->b_v1, t->w_v2, t->b_v2,
                       priors, values, t->HW, t->O);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int tower_forward(Tower *t, const int8_t *planes_dev, int B, float *priors_dev, float *values_dev, hipStream_t st, std::string &err)
{
    if (!t->ready) { err = "weights not loaded"; return -1; }
    if (B < 1 || B > t->maxB) { err = "batch exceeds the tower's activation buffers"; return -1; }
    t->evs_used = 0;
    int rc = t->dtype == 1 ? forward_t<bf16_t>(t, planes_dev, B, priors_dev, values_dev, st)
                           : forward_t<float>(t, planes_dev, B, priors_dev, values_dev, st);
    if (rc) err = std::string("kernel launch failed: ") + hipGetErrorString(hipGetLastError());
    return rc;
}

void tower_set_profile(Tower *t, bool on)
{
    t->profile = on;
    if (on && t->evs.empty()) {
        t->evs.resize((size_t)4 * t->NB);
        for (auto &e : t->evs) hipEventCreate(&e);
    }
}

// call after the launch stream has been synchronised: folds the last forward's event pairs
void tower_collect_profile(Tower *t)
{
    for (int i = 0; i + 1 < t->evs_used; i += 2) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t->evs[i], t->evs[i + 1]) == hipSuccess) { t->conv_ms += ms; t->conv_launches++; }
    }
    t->evs_used = 0;
}

void tower_profile_read(Tower *t, double *conv_ms, unsigned long long *launches, bool reset)
{
    *conv_ms = t->conv_ms; *launches = t->conv_launches;
    if (reset) { t->conv_ms = 0; t->conv_launches = 0; }
}
