// Convolution kernels of the tower (device code only); included by az_tower.hip and by tools/conv_bench.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef unsigned short bf16_t;

#define CIN0 32  // input planes (17) padded to one K chunk

__device__ __forceinline__ bf16_t f2bf(float f)
{
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((unsigned int)b) << 16); }

// two floats -> packed bf16 pair (round to nearest even; one v_cvt_pk_bf16_f32)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    f32x2_t v = {lo, hi};
    bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(unsigned, r);
}

// ---------------------------------------------------------------------------
// The 16-bit storage format of activations and weights is a template parameter F of every kernel below:
//   F = 0  bfloat16 (8 significant bits, float32's exponent range)
//   F = 1  IEEE half precision (11 significant bits, largest finite value 65504)
// Both feed the same MFMA shape at the same rate and move the same bytes; half precision rounds 8 times finer, which is
// what a 10- or 20-block tower needs to keep its value output within 1e-2 of float32.  Activations are clamped to the
// format's finite range on conversion (a bf16 tower cannot overflow; a half-precision one must not turn a large
// activation into an infinity).  The element type in memory is a plain 16-bit word (bf16_t) for both.
// ---------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

template <int F> __device__ __forceinline__ bf16_t f2h(float f)
{
    if constexpr (F == 0) return f2bf(f);
    else return __builtin_bit_cast(bf16_t, (_Float16)__builtin_amdgcn_fmed3f(f, -65504.0f, 65504.0f));
}
template <int F> __device__ __forceinline__ float h2f(bf16_t h)
{
    if constexpr (F == 0) return bf2f(h);
    else return (float)__builtin_bit_cast(_Float16, h);
}
template <int F> __device__ __forceinline__ unsigned pack2(float lo, float hi)
{
    if constexpr (F == 0) return pack_bf16(lo, hi);
    else {
        f16x2_t r = {(_Float16)__builtin_amdgcn_fmed3f(lo, -65504.0f, 65504.0f), (_Float16)__builtin_amdgcn_fmed3f(hi, -65504.0f, 65504.0f)};
        return __builtin_bit_cast(unsigned, r);
    }
}
// the two halves of a packed pair as floats
template <int F> __device__ __forceinline__ float lo2f(unsigned u)
{
    if constexpr (F == 0) return __uint_as_float(u << 16);
    else return (float)__builtin_bit_cast(f16x2_t, u)[0];
}
template <int F> __device__ __forceinline__ float hi2f(unsigned u)
{
    if constexpr (F == 0) return __uint_as_float(u & 0xffff0000u);
    else return (float)__builtin_bit_cast(f16x2_t, u)[1];
}
// D += A * B on the matrix cores, 16x16x32 and 32x32x16, 16-bit operands (eight per lane), float32 accumulators
template <int F> __device__ __forceinline__ f32x4_t mfma16(bf16x8 a, bf16x8 b, f32x4_t c)
{
    if constexpr (F == 0) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <int F> __device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c)
{
    if constexpr (F == 0) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// the same 16x16x32 product with the accumulator tied as destination (see the slab kernel)
#define AZ_MFMA16_INPLACE(F, acc, a, b)                                                                              \
    do {                                                                                                            \
        if constexpr ((F) == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); \
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));                    \
    } while (0)

// ---------------------------------------------------------------------------
// int8 planes (B,17,H,W) -> NHWC with halo, CIN0 channels (17 real + zeros)
// ---------------------------------------------------------------------------
template <typename T, int F = 0>
__global__ void k_planes_to_nhwc(const int8_t *planes, T *X0, int H, int W, const int32_t *__restrict__ rows,
                                 const int32_t *__restrict__ cnt)
{
    if (cnt && (int)blockIdx.x >= *cnt) return;  // batch size kept on the device: the grid covers the capacity
    // board b of the batch is row rows[b] of `planes` (rows == nullptr: row b).  One thread per board point: 17 plane
    // bytes in (coalesced across the points of a plane), CIN0 channels out as 16-byte stores.
    const int b = blockIdx.x, HW = H * W, PP = (H + 2) * (W + 2);
    const int8_t *src = planes + (size_t)(rows ? rows[b] : b) * 17 * HW;
    for (int pix = blockIdx.y * blockDim.x + threadIdx.x; pix < HW; pix += blockDim.x * gridDim.y) {
        const int y = pix / W, x = pix - y * W;
        T v[CIN0];
#pragma unroll
        for (int c = 0; c < CIN0; c++) {
            const float f = c < 17 ? (float)src[c * HW + pix] : 0.0f;
            if constexpr (sizeof(T) == 2) v[c] = f2h<F>(f); else v[c] = f;
        }
        T *dst = X0 + ((size_t)b * PP + (y + 1) * (W + 2) + (x + 1)) * CIN0;
        constexpr int PER = 16 / (int)sizeof(T);
#pragma unroll
        for (int q = 0; q < CIN0 / PER; q++) *reinterpret_cast<uint4 *>(dst + q * PER) = *reinterpret_cast<const uint4 *>(&v[q * PER]);
    }
}

// ---------------------------------------------------------------------------
// bf16 implicit-GEMM 3x3 convolution on MFMA.
// Workgroup tile 128 pixels x 128 output channels, 4 waves as 2x2, each wave a
// 64x64 tile = 2x2 MFMA 32x32 accumulators; K step = one tap x KC channels.
// ---------------------------------------------------------------------------
template <int KC, bool RESID, int F = 0>
__global__ __launch_bounds__(256) void k_conv3x3_bf16(const bf16_t *__restrict__ X, const bf16_t *__restrict__ Wt,
                                                      const float *__restrict__ bias, const bf16_t *R,
                                                      bf16_t *Y, int M, int HW, int Wd, int PP, int CIN, int COUT)
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
                        acc[i][j] = mfma32<F>(af[i], bf[j], acc[i][j]);
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
                    if (RESID) v += h2f<F>(R[o]);
                    Y[o] = f2h<F>(fmaxf(v, 0.0f));
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

template <bool RESID, int F = 0>
__global__ __launch_bounds__(512) void k_conv3x3_c256_bf16(const bf16_t *__restrict__ X, const bf16_t *__restrict__ Wt,
                                                           const float *__restrict__ bias, const bf16_t *R,
                                                           bf16_t *Y, int M, int HW, int Wd, int PP)
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
                    acc[i][j] = mfma16<F>(wf[i], xf[j], acc[i][j]);
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
            pk.x = pack2<F>(a[0] + b4.x, a[1] + b4.y);
            pk.y = pack2<F>(a[2] + b4.z, a[3] + b4.w);
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
                float lo = lo2f<F>(vv[k]), hi = hi2f<F>(vv[k]);
                if (RESID) { lo += lo2f<F>(rr[k]); hi += hi2f<F>(rr[k]); }
                out[k] = pack2<F>(fmaxf(lo, 0.f), fmaxf(hi, 0.f));
            }
            *reinterpret_cast<uint4 *>(Y + o) = make_uint4(out[0], out[1], out[2], out[3]);
        }
    }
}

// bias -> bf16 -> per-wave LDS transpose -> 16-byte rows (+ residual, ReLU) -> NHWC stores.
// All LDS reads of the K loop must be complete (barrier) before the call.
// R and Y may be the same buffer (the tower in groups of boards writes a block's output over its residual input): every
// output element depends on the residual at that element only, and neither pointer is declared restrict in any bf16
// convolution kernel, so the compiler keeps each element's residual load ahead of its store.
template <bool RESID, int NI, int F = 0>
__device__ __forceinline__ void conv_epilogue(unsigned char *smem, const int *rowoff, f32x4 (&acc)[NI][8], const float *__restrict__ bias,
                                              const bf16_t *R, bf16_t *Y, int M, int m0, int wave, int lane,
                                              int wr, int cbase)
{
    constexpr int C = 256;
    constexpr int CPR = NI * 2;       // 16-byte pieces (8 channels) in one point's row of this wave
    constexpr int RPP = 64 / CPR;     // points the wave stores per pass
    constexpr int NPASS = 128 / RPP;
    unsigned char *reg = smem + wave * 18432;
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const float4 b4 = *reinterpret_cast<const float4 *>(bias + cbase + i * 16 + (lane >> 4) * 4);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const f32x4 a = acc[i][j];
            uint2 pk;
            pk.x = pack2<F>(a[0] + b4.x, a[1] + b4.y);
            pk.y = pack2<F>(a[2] + b4.z, a[3] + b4.w);
            *reinterpret_cast<uint2 *>(reg + (j * 16 + (lane & 15)) * 144 + (i * 16 + (lane >> 4) * 4) * 2) = pk;
        }
    }
    const int erow = lane / CPR, echunk = lane % CPR;
    uint4 rres[NPASS];
    if (RESID) {
#pragma unroll
        for (int it = 0; it < NPASS; it++) {
            const int trow = wr * 128 + it * RPP + erow;
            rres[it] = make_uint4(0, 0, 0, 0);
            if (m0 + trow < M) rres[it] = *reinterpret_cast<const uint4 *>(R + (size_t)rowoff[trow] * C + cbase + echunk * 8);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NPASS; it++) {
        const int row = it * RPP + erow;
        const int trow = wr * 128 + row;
        if (m0 + trow < M) {
            uint4 v = *reinterpret_cast<const uint4 *>(reg + row * 144 + echunk * 16);
            const size_t o = (size_t)rowoff[trow] * C + cbase + echunk * 8;
            unsigned vv[4] = {v.x, v.y, v.z, v.w};
            unsigned rr[4] = {0, 0, 0, 0};
            if (RESID) { rr[0] = rres[it].x; rr[1] = rres[it].y; rr[2] = rres[it].z; rr[3] = rres[it].w; }
            unsigned out[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float lo = lo2f<F>(vv[k]), hi = hi2f<F>(vv[k]);
                if (RESID) { lo += lo2f<F>(rr[k]); hi += hi2f<F>(rr[k]); }
                out[k] = pack2<F>(fmaxf(lo, 0.f), fmaxf(hi, 0.f));
            }
            *reinterpret_cast<uint4 *>(Y + o) = make_uint4(out[0], out[1], out[2], out[3]);
        }
    }
}

// ---------------------------------------------------------------------------
// Raw workgroup barrier for the ping-pong schedule of the slab kernel below (no implicit waitcnt, so
// LDS-DMA stays in flight across it).  (tools/conv_ablation_kernels.h is this file's harness copy with the ablation
// and shader-clock-stamp switches that the timing studies were made with; nothing here is conditional.)
// ---------------------------------------------------------------------------
#define PP_BAR() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

// ---------------------------------------------------------------------------
// Slab variant (the one the tower uses when the tile's pixel span fits): the activation operand
// is NOT re-fetched per tap.  For each 32-channel chunk the workgroup loads once, by LDS-DMA, the
// "slab" of all padded positions its 256 pixels touch through any of the 9 taps (a contiguous range
// of at most NSP*128 padded positions, 64 B each) and the 9 taps read it at shifted rows; only the
// weight half-tiles stream per half-step (ring of 4 x 16 KiB, three in flight).  The K loop is cut into
// half-steps (one tap x 32 channels: 12 fragment reads, 32 MFMAs per wave) and the two 4-wave groups
// (pixel halves wr = 0 / 1, which share the SIMDs pairwise) run one barrier interval apart: while one
// group issues its MFMAs (and its share of the LDS-DMA) the other reads its next fragments from LDS, so
// the matrix pipe of every SIMD always has a wave to serve.  DMA completion is tracked with counted
// s_waitcnt vmcnt(N), never 0 inside the loop.  L2->LDS traffic per
// tile drops from 2.36 MB to 1.37 MB.  Slot s of a 64-byte row r holds k-chunk s ^ (((r >> 2) & 1) << 1),
// which keeps ds_read_b128 conflict-free for any alignment of 16 consecutive rows.
// ---------------------------------------------------------------------------
#define V5_SLAB0 65536
#define V5_LDS_BYTES (147456 + 2048)
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Workgroups are handed to the 8 XCDs round-robin (workgroup i -> XCD i % 8), each XCD with its own L2.  Tiles that
// are neighbours in memory share their halo rows, so workgroup i takes tile (i % 8) * ceil-share + i / 8: every XCD
// walks one contiguous range of tiles and the halo rows of a tile are already in that XCD's L2 (bijective for any
// workgroup count).  Placement is only a matter of speed, never of correctness.
__device__ __forceinline__ int xcd_tile(int wg, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
}

// CW = output channels per wave (64, 32 or 16): a workgroup covers 4 * CW channels of its 256 points and blockIdx.y
// picks which, so a batch with few tiles can still occupy the chip (each workgroup then re-reads the activation slab
// but only its share of the weights).
template <bool RESID, int NSP, int CIN = 256, int CW = 64, int F = 0>
__global__ __launch_bounds__(512) void k_conv3x3_c256_bf16_slab(const bf16_t *__restrict__ X, const bf16_t *__restrict__ Wt,
                                                                const float *__restrict__ bias, const bf16_t *R,
                                                                bf16_t *Y, int M, int HW, int Wd, int PP, int npos,
                                                                const int32_t *__restrict__ cnt)
{
    constexpr int NCC = CIN / 32, SLAB_BYTES = NSP * 128 * 64;  // NCC 32-channel input chunks; 256 output channels in all
    constexpr int NI = CW / 16;            // 16-channel MFMA blocks per wave
    constexpr int WROWS = 4 * CW;          // weight rows (output channels) of this workgroup
    constexpr int WB = CW == 64 ? 2 : 1;   // LDS-DMA loads per wave and weight batch (16 rows each)
    static_assert(CW == 64 || CW == 32 || CW == 16, "channels per wave");
    const int cout0 = blockIdx.y * WROWS;
    // with the number of boards kept on the device the grid covers the capacity M / HW: the first ceil(M / 256)
    // workgroups (still round-robin over the XCDs) share the live tiles, the others have nothing to do
    int nwg = gridDim.x;
    if (cnt) {
        const int nb = *cnt;
        M = nb * HW; npos = nb * PP;
        nwg = (M + 255) >> 8;
        if ((int)blockIdx.x >= nwg) return;
    }
    const int tile = xcd_tile(blockIdx.x, nwg);
    static_assert(V5_SLAB0 + 2 * SLAB_BYTES <= 147456, "slab buffers overflow LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[V5_LDS_BYTES];
    int *rowoff = reinterpret_cast<int *>(smem + 147456);
    int *srow = reinterpret_cast<int *>(smem + 147456 + 1024);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int m0 = tile * 256, W2 = Wd + 2;
    if (tid < 256) {
        int m = min(m0 + tid, M - 1);
        int b = m / HW, pix = m - b * HW, y = pix / Wd, x = pix - y * Wd;
        rowoff[tid] = b * PP + (y + 1) * W2 + (x + 1);
    }
    __syncthreads();
    const int pbase = rowoff[0] - (W2 + 1);
    if (tid < 256) srow[tid] = (rowoff[tid] - pbase) * 64;  // byte offset of the pixel's own row + (W2+1) rows
    // weight fill mapping: wave w fills rows w*32 + q*16 + (lane>>2), slot lane&3
    const int frow = lane >> 2, fslot = lane & 3;
    int woff[WB];
#pragma unroll
    for (int q = 0; q < WB; q++) {
        // CW = 16 has 64 rows for 8 waves: waves 4-7 repeat the rows of waves 0-3 (same bytes to the same place), which
        // keeps the number of loads per wave, and with it the vmcnt bookkeeping, the same for every wave
        const int r = (wave * 16 * WB + q * 16 + frow) % WROWS;
        woff[q] = (cout0 + r) * 9 * CIN + ((fslot ^ (((r >> 2) & 1) << 1)) << 3);
    }
    // slab fill mapping: wave w fills pieces w + 8*i (16 rows each)
    int soff[NSP];
#pragma unroll
    for (int i = 0; i < NSP; i++) {
        const int sr = (wave + 8 * i) * 16 + frow;
        const int pos = min(pbase + sr, npos - 1);
        soff[i] = pos * CIN + ((fslot ^ (((sr >> 2) & 1) << 1)) << 3);
    }
    unsigned char *const wfill = smem + ((wave * 16 * WB) % WROWS) * 64;
    const int wrdo = (wc * CW + (lane & 15)) * 64 + ((((lane >> 4)) ^ ((((lane & 15) >> 2) & 1) << 1)) << 4);
    __syncthreads();
    int xrow[8];  // byte offset of each fragment's pixel row inside the slab (+ k-chunk), before the tap shift
#pragma unroll
    for (int j = 0; j < 8; j++) xrow[j] = srow[wr * 128 + j * 16 + (lane & 15)] + ((lane >> 4) << 4);

    f32x4 acc[NI][8];
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 wf[NI], xf[8];

    int opaque0 = 0;  // re-defined by an empty asm every chunk so that per-tap addresses are not hoisted into VGPRs
    // (cc, tap) = 32-channel chunk and filter tap of half-step h = 9*cc + tap; ring slot of its weights = h & 3
    auto issue_w = [&](int cc, int tap) {
        const int wofs = tap * CIN + cc * 32 + opaque0;
        unsigned char *dst = wfill + ((cc + tap) & 3) * 16384;
#pragma unroll
        for (int q = 0; q < WB; q++) glds16(Wt + woff[q] + wofs, dst + q * 1024);
    };
    auto issue_slab = [&](int cc) {
        unsigned char *dst = smem + V5_SLAB0 + (cc & 1) * SLAB_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < NSP; i++) glds16(X + soff[i] + cc * 32, dst + i * 8192);
    };
    auto rd_frags = [&](int cc, int tap) {
        const int ty = tap / 3, tx = tap - ty * 3;
        const int tshift = ((ty - 1) * W2 + (tx - 1)) * 64 + opaque0;  // relative to the pixel's own row
        const unsigned char *wb = smem + ((cc + tap) & 3) * 16384 + wrdo;
        const unsigned char *sb = smem + V5_SLAB0 + (cc & 1) * SLAB_BYTES;
#pragma unroll
        for (int i = 0; i < NI; i++) wf[i] = *reinterpret_cast<const bf16x8 *>(wb + i * 1024);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int a = xrow[j] + tshift;  // row*64 + chunk*16
            xf[j] = *reinterpret_cast<const bf16x8 *>(sb + (a ^ ((a & 256) >> 3)));  // slot ^= 2 when row bit 2 is set
        }
    };
    auto mfma_rows = [&](int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; i++)
#pragma unroll
            for (int j = 0; j < 8; j++)
                // accumulate in place: tying the destination to the C operand keeps the register allocator from
                // renaming the 128 accumulator registers from tap to tap (which costs spills inside the loop)
                AZ_MFMA16_INPLACE(F, acc[i][j], wf[i], xf[j]);
    };
    // One MFMA interval: 32 MFMAs with this wave's share of the LDS-DMA slipped in after the first rows.
    // dw = how many half-steps ahead the weights are fetched (3 for group 0, 2 for group 1, which issues
    // one interval later); slab_tap = the tap at which the next chunk's slab is issued.
    auto mfma_interval = [&](int cc, int tap, int dw, int slab_tap) {
        __builtin_amdgcn_s_setprio(1);
        mfma_rows(0, 1);
        __builtin_amdgcn_sched_barrier(0);
        const int tn = tap + dw;
        if (tn < 9) issue_w(cc, tn);
        else if (cc < NCC - 1) issue_w(cc + 1, tn - 9);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(1, NI < 2 ? NI : 2);
        __builtin_amdgcn_sched_barrier(0);
        if (tap == slab_tap && cc < NCC - 1) issue_slab(cc + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(NI < 2 ? NI : 2, NI);
        __builtin_amdgcn_s_setprio(0);
    };
    // vmcnt bookkeeping (per wave, in issue order): before half-step h+1 is read its weights must have
    // landed; `ahead` younger weight batches (2 glds each) may stay in flight, plus the NSP slab pieces
    // while they are younger than W(h+1).
    auto retire_g0 = [&](int cc, int tap) {  // end of group 0's MFMA interval of (cc, tap); issued up to W(h+3)
        if (cc < NCC - 1) { if (tap < 3) wait_vmcnt<2 * WB + NSP>(); else wait_vmcnt<2 * WB>(); }
        else if (tap <= 5) wait_vmcnt<2 * WB>();
        else if (tap == 6) wait_vmcnt<WB>();
        else wait_vmcnt<0>();
    };
    auto retire_g1 = [&](int cc, int tap) {  // end of group 1's read interval of (cc, tap); issued up to W(h+2)
        if (cc < NCC - 1) { if (tap == 1 || tap == 2) wait_vmcnt<WB + NSP>(); else wait_vmcnt<WB>(); }
        else if (tap <= 6) wait_vmcnt<WB>();
        else wait_vmcnt<0>();
    };
#define SB() __builtin_amdgcn_sched_barrier(0)

    issue_slab(0); issue_w(0, 0); issue_w(0, 1); issue_w(0, 2);
    wait_vmcnt<2 * WB>();
    PP_BAR();
    if (wr == 0) {
        for (int cc = 0; cc < NCC; cc++) {
            asm volatile("" : "+s"(opaque0));
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                rd_frags(cc, tap);
                SB(); PP_BAR(); SB();
                mfma_interval(cc, tap, 3, 0);
                retire_g0(cc, tap);
                SB(); PP_BAR(); SB();
            }
        }
        PP_BAR();
    } else {
        for (int cc = 0; cc < NCC; cc++) {
            asm volatile("" : "+s"(opaque0));
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                // MFMAs of the previous half-step (its fragments were read one interval ago)
                if (tap > 0) mfma_interval(cc, tap - 1, 3, 0);
                else if (cc > 0) mfma_interval(cc - 1, 8, 3, 0);
                SB(); PP_BAR(); SB();
                rd_frags(cc, tap);
                retire_g1(cc, tap);
                SB(); PP_BAR(); SB();
            }
        }
        mfma_interval(NCC - 1, 8, 3, 0);
        PP_BAR();
    }
#undef SB
    // the compiler does not know that the asm statements above are MFMAs, so it cannot insert the wait
    // states an MFMA result needs before a vector ALU read: cover the longest case by hand
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    // (a register-swap epilogue without the LDS round trip measured 1-3 % slower in the self-play bench)
    __syncthreads();
    conv_epilogue<RESID, NI, F>(smem, rowoff, acc, bias, R, Y, M, m0, wave, lane, wr, cout0 + wc * CW);
}

// ---------------------------------------------------------------------------
// Two-workgroups-per-CU form of the slab kernel.  A workgroup of 4 waves owns 256 points x 128 output channels (blockIdx
// picks the channel half), needs 74 KB of LDS (slab of 24 pieces of 16 positions: 19x19 boards) or exactly 80 KB (27
// pieces = 432 positions: 15x15, 9x9, 8x8 boards, whose tiles span up to 426), and two of them share a compute unit: each
// SIMD then holds one wave of either workgroup, and while one workgroup sits in its prologue (first DMA), at a barrier or
// in its epilogue (transpose, residual, stores - 11 to 15 % of a tile's life in the one-workgroup kernel, where nothing
// else is resident to cover it) the other one's wave keeps the matrix pipe busy.  No hand-made ping-pong: the two
// workgroups are not synchronised with each other.  Per half-step (one tap x 32 input channels) a wave waits for its
// share of the weights' LDS-DMA (counted vmcnt), passes ONE raw barrier, issues the DMA two half-steps ahead into the
// ring slot that barrier has just freed, reads 4 + 8 fragments and issues 32 MFMAs.  The slab is shared by the two
// channel halves of a tile only through L2 (both are placed on the same XCD, next to each other).
// ---------------------------------------------------------------------------
#define V6_RING 0
#define V6_SLAB0 24576
// NPW = slab pieces (16 positions x 64 B) each wave fetches per channel chunk: 6 (24 pieces) or 7 (27 distinct pieces; the
// 28th load repeats the 27th, so that every wave issues the same number of loads and the vmcnt bookkeeping stays uniform)
template <bool RESID, int NPW = 6, int F = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_conv3x3_c256_bf16_duo(const bf16_t *__restrict__ X, const bf16_t *__restrict__ Wt, const float *__restrict__ bias,
                             const bf16_t *R, bf16_t *Y, int M, int HW, int Wd, int PP, int npos,
                             const int32_t *__restrict__ cnt)
{
    constexpr int CIN = 256, NCC = 8, NPIECES = NPW == 6 ? 24 : 27, SLAB_BYTES = NPIECES * 1024;
    constexpr int V6_TABLES = V6_SLAB0 + 2 * SLAB_BYTES, V6_LDS_BYTES = V6_TABLES + 2048;
    static_assert(NPW == 6 || NPW == 7, "24 or 27 slab pieces");
    static_assert(V6_LDS_BYTES <= 81920 && V6_TABLES >= 4 * 18432, "two workgroups per 160 KB; the epilogue's staging fits below the tables");
    int nwg = gridDim.x;
    if (cnt) {
        const int nb = *cnt;
        M = nb * HW; npos = nb * PP;
        nwg = 2 * ((M + 255) >> 8);
        if ((int)blockIdx.x >= nwg) return;
    }
    const int L = xcd_tile(blockIdx.x, nwg);
    const int tile = L >> 1, cout0 = (L & 1) * 128;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[V6_LDS_BYTES];
    int *rowoff = reinterpret_cast<int *>(smem + V6_TABLES);
    int *srow = reinterpret_cast<int *>(smem + V6_TABLES + 1024);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = tile * 256, W2 = Wd + 2;
    {
        int m = min(m0 + tid, M - 1);
        int b = m / HW, pix = m - b * HW, y = pix / Wd, x = pix - y * Wd;
        rowoff[tid] = b * PP + (y + 1) * W2 + (x + 1);
    }
    __syncthreads();
    const int pbase = rowoff[0] - (W2 + 1);
    srow[tid] = (rowoff[tid] - pbase) * 64;
    const int frow = lane >> 2, fslot = lane & 3;
    int woff[2];   // weights: wave w fills rows w*32 + q*16 + (lane >> 2) of the 128-row stage
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int r = wave * 32 + q * 16 + frow;
        woff[q] = (cout0 + r) * 9 * CIN + ((fslot ^ (((r >> 2) & 1) << 1)) << 3);
    }
    int soff[NPW];   // slab: wave w fills the 16-row pieces w + 4*i
    const int last_piece = min(wave + 4 * (NPW - 1), NPIECES - 1);
#pragma unroll
    for (int i = 0; i < NPW; i++) {
        const int sr = (i == NPW - 1 ? last_piece : wave + 4 * i) * 16 + frow;
        const int pos = min(pbase + sr, npos - 1);
        soff[i] = pos * CIN + ((fslot ^ (((sr >> 2) & 1) << 1)) << 3);
    }
    const int wrdo = (wc * 64 + (lane & 15)) * 64 + ((((lane >> 4)) ^ ((((lane & 15) >> 2) & 1) << 1)) << 4);
    __syncthreads();
    int xrow[8];
#pragma unroll
    for (int j = 0; j < 8; j++) xrow[j] = srow[wr * 128 + j * 16 + (lane & 15)] + ((lane >> 4) << 4);

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 wf[4], xf[8];
    int opaque0 = 0;

    auto issue_w = [&](int cc, int tap) {   // weights of half-step h = 9*cc + tap -> ring slot h % 3 = tap % 3
        const int wofs = tap * CIN + cc * 32 + opaque0;
        unsigned char *dst = smem + V6_RING + (tap % 3) * 8192 + wave * 2048;
#pragma unroll
        for (int q = 0; q < 2; q++) glds16(Wt + woff[q] + wofs, dst + q * 1024);
    };
    auto issue_slab = [&](int cc) {
        unsigned char *dst = smem + V6_SLAB0 + (cc & 1) * SLAB_BYTES;
#pragma unroll
        for (int i = 0; i < NPW; i++) glds16(X + soff[i] + cc * 32, dst + (i == NPW - 1 ? last_piece : wave + 4 * i) * 1024);
    };

    issue_slab(0); issue_w(0, 0); issue_w(0, 1);
    for (int cc = 0; cc < NCC; cc++) {
        asm volatile("" : "+s"(opaque0));
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            // this wave's share of W(h), h = 9*cc + tap, (and of the chunk's slab) has landed; younger loads may stay in flight: W(h+1), and
            // the NPW slab pieces issued one or two half-steps ago
            if (cc == NCC - 1 && tap == 8) wait_vmcnt<0>();
            else if (cc < NCC - 1 && (tap == 1 || tap == 2)) wait_vmcnt<2 + NPW>();
            else wait_vmcnt<2>();
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");  // everybody's share has landed, and everybody is done reading half-step h-1
            __builtin_amdgcn_sched_barrier(0);
            {   // fragment reads first: their LDS latency runs while the DMA for two half-steps ahead is issued
                const int ty = tap / 3, tx = tap - ty * 3;
                const int tshift = ((ty - 1) * W2 + (tx - 1)) * 64 + opaque0;
                const unsigned char *wb = smem + V6_RING + (tap % 3) * 8192 + wrdo;
                const unsigned char *sb = smem + V6_SLAB0 + (cc & 1) * SLAB_BYTES;
#pragma unroll
                for (int i = 0; i < 4; i++) wf[i] = *reinterpret_cast<const bf16x8 *>(wb + i * 1024);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int a = xrow[j] + tshift;
                    xf[j] = *reinterpret_cast<const bf16x8 *>(sb + (a ^ ((a & 256) >> 3)));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 2 < 9) issue_w(cc, tap + 2);
            else if (cc < NCC - 1) issue_w(cc + 1, tap + 2 - 9);
            if (tap == 0 && cc < NCC - 1) issue_slab(cc + 1);
            __builtin_amdgcn_sched_barrier(0);
            // point-block major: the first MFMAs need only the weights and the first point fragment
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    AZ_MFMA16_INPLACE(F, acc[i][j], wf[i], xf[j]);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __syncthreads();
    conv_epilogue<RESID, 4, F>(smem, rowoff, acc, bias, R, Y, M, m0, wave, lane, wr, cout0 + wc * 64);
}

// largest number of padded positions (incl. the one-row-plus-one halo on both sides) that a tile of
// 256 consecutive pixels can span; decides whether / with which NSP the slab kernel applies
static inline int conv_slab_span(int HW, int Wd, int PP)
{
    const int W2 = Wd + 2;
    int worst = 0;
    for (int r = 0; r < HW; r++) {
        auto pos = [&](long m) { long b = m / HW; int pix = (int)(m - b * HW), y = pix / Wd, x = pix - y * Wd; return b * PP + (y + 1) * W2 + (x + 1); };
        int span = (int)(pos(r + 255) - pos(r)) + 1 + 2 * (W2 + 1);
        if (span > worst) worst = span;
    }
    return worst;
}

// stem: 32 (zero padded) input channels -> 256, one 32-channel chunk, same kernel
template <int F = 0>
static inline bool launch_v5_stem(const bf16_t *X, const bf16_t *Wt, const float *bias, bf16_t *Y, int M, int HW, int Wd, int PP,
                                  hipStream_t st = 0, const int32_t *cnt = nullptr);

// How many ways the 256 output channels of a tile are split over workgroups (blockIdx.y): with fewer tiles than
// compute units a whole-tile workgroup leaves most of the chip idle, so the tiles are cut along the channel axis.
// A workgroup's time is set mostly by its 72 dependent K steps, not by its MFMAs (whole tile 52 us, half 41, quarter 37
// on 19x19 boards), so splitting pays only while all pieces run at once.
static inline int conv_channel_split(int M, int compute_units = 256)
{
    const int tiles = (M + 255) / 256;
    if (tiles * 4 <= compute_units) return 4;
    if (tiles * 2 <= compute_units) return 2;
    return 1;
}

template <bool RESID, int NSP, int F = 0>
static inline void launch_v5_cw(int cs, dim3 g, dim3 blk, hipStream_t st, const bf16_t *X, const bf16_t *Wt, const float *bias,
                                const bf16_t *R, bf16_t *Y, int M, int HW, int Wd, int PP, int npos, const int32_t *cnt)
{
    g.y = cs;
    if (cs == 4) hipLaunchKernelGGL((k_conv3x3_c256_bf16_slab<RESID, NSP, 256, 16, F>), g, blk, 0, st, X, Wt, bias, R, Y, M, HW, Wd, PP, npos, cnt);
    else if (cs == 2) hipLaunchKernelGGL((k_conv3x3_c256_bf16_slab<RESID, NSP, 256, 32, F>), g, blk, 0, st, X, Wt, bias, R, Y, M, HW, Wd, PP, npos, cnt);
    else hipLaunchKernelGGL((k_conv3x3_c256_bf16_slab<RESID, NSP, 256, 64, F>), g, blk, 0, st, X, Wt, bias, R, Y, M, HW, Wd, PP, npos, cnt);
}

// the two-workgroups-per-CU form: boards whose slab fits 384 or 432 positions, batches that fill the chip
template <int F = 0>
static inline bool launch_v6(const bf16_t *X, const bf16_t *Wt, const float *bias, const bf16_t *R, bf16_t *Y, int M, int HW, int Wd,
                             int PP, hipStream_t st = 0, const int32_t *cnt = nullptr)
{
    const int span = conv_slab_span(HW, Wd, PP);
    if (span > 27 * 16) return false;
    const int npos = (M / HW) * PP;
    dim3 g(2 * ((M + 255) / 256)), blk(256);
    if (span <= 24 * 16) {
        if (R) hipLaunchKernelGGL((k_conv3x3_c256_bf16_duo<true, 6, F>), g, blk, 0, st, X, Wt, bias, R, Y, M, HW, Wd, PP, npos, cnt);
        else hipLaunchKernelGGL((k_conv3x3_c256_bf16_duo<false, 6, F>), g, blk, 0, st, X, Wt, bias, R, Y, M, HW, Wd, PP, npos, cnt);
    } else {
        if (R) hipLaunchKernelGGL((k_conv3x3_c256_bf16_duo<true, 7, F>), g, blk, 0, st, X, Wt, bias, R, Y, M, HW, Wd, PP, npos, cnt);
        else hipLaunchKernelGGL((k_conv3x3_c256_bf16_duo<false, 7, F>), g, blk, 0, st, X, Wt, bias, R, Y, M, HW, Wd, PP, npos, cnt);
    }
    return true;
}

// cs: channel split (1, 2 or 4), 0 = choose from the batch size
template <int F = 0>
static inline bool launch_v5(const bf16_t *X, const bf16_t *Wt, const float *bias, const bf16_t *R, bf16_t *Y, int M, int HW, int Wd,
                             int PP, hipStream_t st = 0, const int32_t *cnt = nullptr, int cs = 0)
{
    const int span = conv_slab_span(HW, Wd, PP);
    const int npos = (M / HW) * PP;
    dim3 g((M + 255) / 256), blk(512);
    if (cs == 0) cs = conv_channel_split(M);
    if (span <= 3 * 128) {
        if (R) launch_v5_cw<true, 3, F>(cs, g, blk, st, X, Wt, bias, R, Y, M, HW, Wd, PP, npos, cnt);
        else launch_v5_cw<false, 3, F>(cs, g, blk, st, X, Wt, bias, R, Y, M, HW, Wd, PP, npos, cnt);
        return true;
    }
    if (span <= 4 * 128) {
        if (R) launch_v5_cw<true, 4, F>(cs, g, blk, st, X, Wt, bias, R, Y, M, HW, Wd, PP, npos, cnt);
        else launch_v5_cw<false, 4, F>(cs, g, blk, st, X, Wt, bias, R, Y, M, HW, Wd, PP, npos, cnt);
        return true;
    }
    return false;
}

template <int F>
static inline bool launch_v5_stem(const bf16_t *X, const bf16_t *Wt, const float *bias, bf16_t *Y, int M, int HW, int Wd, int PP,
                                  hipStream_t st, const int32_t *cnt)
{
    const int span = conv_slab_span(HW, Wd, PP);
    const int npos = (M / HW) * PP;
    dim3 g((M + 255) / 256), blk(512);
    const bf16_t *R = nullptr;
    if (span <= 3 * 128) { hipLaunchKernelGGL((k_conv3x3_c256_bf16_slab<false, 3, 32, 64, F>), g, blk, 0, st, X, Wt, bias, R, Y, M, HW, Wd, PP, npos, cnt); return true; }
    if (span <= 4 * 128) { hipLaunchKernelGGL((k_conv3x3_c256_bf16_slab<false, 4, 32, 64, F>), g, blk, 0, st, X, Wt, bias, R, Y, M, HW, Wd, PP, npos, cnt); return true; }
    return false;
}

// ---------------------------------------------------------------------------
// Small-batch 3x3 convolution (a single position or a handful: Player.think on one game, play_cmd / GTP
// style use).  The slab kernel needs >= 256 tiles of 256 points to fill the chip; one 19x19 board is two.
// Here a workgroup owns 32 board points x 32 or 64 output channels, so one board already spreads over 48-96
// workgroups, and its 4 or 8 waves split the reduction axis (tap x 32-channel steps) between them: operands
// go straight from L2 into registers in MFMA layout (16 B per lane, no LDS staging - there is no reuse
// inside such a small tile worth staging for), several steps of loads in flight per wave, and the
// partial accumulators are folded through LDS before the bias / residual / ReLU epilogue.
// ---------------------------------------------------------------------------
// NW waves per workgroup = ways the reduction axis is split; NI 16-channel blocks per workgroup; UN steps of loads
// in flight per wave.  <8, 2, 9>: one or two boards (72 steps / 8 waves = one batch of loads, lowest latency);
// <4, 4, 3>: up to about 16 boards (fewer, larger workgroups: less operand re-reading).
template <bool RESID, int NW, int NI, int UN, int F = 0>
__global__ __launch_bounds__(64 * NW) void k_conv3x3_bf16_small(const bf16_t *__restrict__ X, const bf16_t *__restrict__ Wt,
                                                                      const float *__restrict__ bias, const bf16_t *R,
                                                                      bf16_t *Y, int M, int HW, int Wd, int PP, int CIN,
                                                                      int COUT)
{
    __shared__ float part[NW][NI * 2][64][4];  // [wave][i*2+j][lane][r]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 32, c0 = blockIdx.y * (16 * NI), W2 = Wd + 2;
    const int fr = lane & 15, fq = lane >> 4;
    // padded position of this lane's point in each of the two 16-point blocks (clamped; stores are masked)
    int pos[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int m = min(m0 + j * 16 + fr, M - 1);
        const int b = m / HW, pix = m - b * HW, y = pix / Wd, x = pix - y * Wd;
        pos[j] = b * PP + (y + 1) * W2 + (x + 1);
    }
    const int nchunk = CIN / 32, nsteps = 9 * nchunk;
    const bf16_t *wrow[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) wrow[i] = Wt + (size_t)(c0 + i * 16 + fr) * 9 * CIN + fq * 8;
    f32x4 acc[NI][2];
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int h0 = wave; h0 < nsteps; h0 += NW * UN) {
        bf16x8 wf[UN][NI], xf[UN][2];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int h = min(h0 + NW * u, nsteps - 1);  // past the end: reload the last step, not used below
            const int tap = h / nchunk, cc = h - tap * nchunk;
            const int ty = tap / 3, tx = tap - ty * 3;
            const int koff = tap * CIN + cc * 32;
            const int xo = ((ty - 1) * W2 + (tx - 1)) * CIN + cc * 32 + fq * 8;
#pragma unroll
            for (int i = 0; i < NI; i++) wf[u][i] = *reinterpret_cast<const bf16x8 *>(wrow[i] + koff);
#pragma unroll
            for (int j = 0; j < 2; j++) xf[u][j] = *reinterpret_cast<const bf16x8 *>(X + (size_t)pos[j] * CIN + xo);
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            if (h0 + NW * u < nsteps) {
#pragma unroll
                for (int i = 0; i < NI; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        acc[i][j] = mfma16<F>(wf[u][i], xf[u][j], acc[i][j]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) *reinterpret_cast<f32x4 *>(&part[wave][i * 2 + j][lane][0]) = acc[i][j];
    __syncthreads();
    // the NI * 2 (channel block i, point block j) pairs are finished by the waves in turn:
    // channels c0 + i*16 + fq*4 .. +3 of point j*16 + fr
    for (int pr = wave; pr < NI * 2; pr += NW) {
        const int i = pr >> 1, j = pr & 1;
        const int ch = c0 + i * 16 + fq * 4;
        const float4 b4 = *reinterpret_cast<const float4 *>(bias + ch);
        f32x4 v = *reinterpret_cast<const f32x4 *>(&part[0][pr][lane][0]);
#pragma unroll
        for (int w = 1; w < NW; w++) v += *reinterpret_cast<const f32x4 *>(&part[w][pr][lane][0]);
        const int pj = j ? pos[1] : pos[0];
        if (m0 + j * 16 + fr < M) {
            const size_t o = (size_t)pj * COUT + ch;
            float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
            if (RESID) {
                const uint2 rr = *reinterpret_cast<const uint2 *>(R + o);
                r0 = lo2f<F>(rr.x); r1 = hi2f<F>(rr.x);
                r2 = lo2f<F>(rr.y); r3 = hi2f<F>(rr.y);
            }
            uint2 pk;
            pk.x = pack2<F>(fmaxf(v[0] + b4.x + r0, 0.f), fmaxf(v[1] + b4.y + r1, 0.f));
            pk.y = pack2<F>(fmaxf(v[2] + b4.z + r2, 0.f), fmaxf(v[3] + b4.w + r3, 0.f));
            *reinterpret_cast<uint2 *>(Y + o) = pk;
        }
    }
}

template <int NW, int NI, int UN, int F = 0>
static inline void launch_small(const bf16_t *X, const bf16_t *Wt, const float *bias, const bf16_t *R, bf16_t *Y, int M, int HW, int Wd,
                                int PP, int cin, int cout, hipStream_t st)
{
    dim3 gs((M + 31) / 32, cout / (16 * NI)), bs(64 * NW);
    if (R) hipLaunchKernelGGL((k_conv3x3_bf16_small<true, NW, NI, UN, F>), gs, bs, 0, st, X, Wt, bias, R, Y, M, HW, Wd, PP, cin, cout);
    else hipLaunchKernelGGL((k_conv3x3_bf16_small<false, NW, NI, UN, F>), gs, bs, 0, st, X, Wt, bias, R, Y, M, HW, Wd, PP, cin, cout);
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
    // double accumulators: a product of two floats is exact in double and the 9 * CIN-term sum then carries no visible
    // rounding, so each layer's output is the correctly rounded float of its exact value (the 1e-5 parity mode has to
    // hold through 41 layers; float accumulation left 1e-5 on the value of a 20-block tower)
    double acc[4][4] = {};
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
                    for (int j = 0; j < 4; j++) acc[i][j] = fma((double)av[i], (double)bv[j], acc[i][j]);
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
            float v = (float)(acc[i][j] + (double)bias[n] + (RESID ? (double)R[o] : 0.0));
            Y[o] = fmaxf(v, 0.0f);
        }
    }
}

