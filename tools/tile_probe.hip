// Probe: what a wave tile of 128 points x 128 channels on one wave per SIMD could gain over the slab kernel's
// 128 x 64 on two waves per SIMD, before any staging cost: LDS fragment reads + MFMAs only (no LDS-DMA, no barriers),
// the same FLOPs per workgroup-sized tile, random operands resident in LDS.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tile_probe.hip -o tools/tile_probe && tools/tile_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define STEPS 72
#define LDS_BYTES 65536

// NW waves per workgroup, wave tile (16*NI) channels x (16*NJ) points; per step NI + NJ fragment reads and NI*NJ MFMAs
template <int NW, int NI, int NJ>
__global__ __launch_bounds__(64 * NW) void k_probe(const uint4 *__restrict__ src, float *__restrict__ out, int reps)
{
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < LDS_BYTES / 16; i += 64 * NW) reinterpret_cast<uint4 *>(smem)[i] = src[i];
    __syncthreads();
    f32x4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // 64-byte rows, lane (l & 15) reads row l & 15 of a 16-row fragment at k-chunk l >> 4, slot swizzled by row bit 2
    const int r = lane & 15, c = lane >> 4;
    const int frag = r * 64 + ((c ^ (((r >> 2) & 1) << 1)) << 4);
    for (int rep = 0; rep < reps; rep++) {
#pragma unroll 2
        for (int h = 0; h < STEPS; h++) {
            const int base = ((h * 5 + wave * 3) & 15) * 1024;   // moves through the buffer like taps move through a slab
            bf16x8 wf[NI], xf[NJ];
#pragma unroll
            for (int i = 0; i < NI; i++) wf[i] = *reinterpret_cast<const bf16x8 *>(smem + ((base + i * 1024 + frag) & (LDS_BYTES - 1)));
#pragma unroll
            for (int j = 0; j < NJ; j++) xf[j] = *reinterpret_cast<const bf16x8 *>(smem + ((base + 32768 + j * 1024 + frag) & (LDS_BYTES - 1)));
#pragma unroll
            for (int i = 0; i < NI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * 64 * NW + tid] = s;
}

// the same wave tile and the same fragment bytes with v_mfma_f32_32x32x16_bf16: (32*NI) channels x (32*NJ) points, per step
// 2*(NI + NJ) fragment reads and 2*NI*NJ MFMAs of twice the FLOPs (half the operand-register reads per FLOP)
template <int NW, int NI, int NJ>
__global__ __launch_bounds__(64 * NW) void k_probe32(const uint4 *__restrict__ src, float *__restrict__ out, int reps)
{
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < LDS_BYTES / 16; i += 64 * NW) reinterpret_cast<uint4 *>(smem)[i] = src[i];
    __syncthreads();
    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[i][j][q] = 0.f;
    const int r = lane & 15, c = lane >> 4;
    const int frag = r * 64 + ((c ^ (((r >> 2) & 1) << 1)) << 4);
    for (int rep = 0; rep < reps; rep++) {
#pragma unroll 2
        for (int h = 0; h < STEPS; h++) {
            const int base = ((h * 5 + wave * 3) & 15) * 1024;
            bf16x8 wf[NI][2], xf[NJ][2];
#pragma unroll
            for (int i = 0; i < NI; i++)
#pragma unroll
                for (int k = 0; k < 2; k++) wf[i][k] = *reinterpret_cast<const bf16x8 *>(smem + ((base + (2 * i + k) * 1024 + frag) & (LDS_BYTES - 1)));
#pragma unroll
            for (int j = 0; j < NJ; j++)
#pragma unroll
                for (int k = 0; k < 2; k++) xf[j][k] = *reinterpret_cast<const bf16x8 *>(smem + ((base + 32768 + (2 * j + k) * 1024 + frag) & (LDS_BYTES - 1)));
#pragma unroll
            for (int k = 0; k < 2; k++)
#pragma unroll
                for (int i = 0; i < NI; i++)
#pragma unroll
                    for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][k], xf[j][k], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int q = 0; q < 16; q++) s += acc[i][j][q];
    out[(size_t)blockIdx.x * 64 * NW + tid] = s;
}

template <int NW, int NI, int NJ>
static void run32(const char *name, const uint4 *src, float *out, int grid, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_probe32<NW, NI, NJ>), dim3(grid), dim3(64 * NW), 0, 0, src, out, reps);
    CK(hipDeviceSynchronize());
    const int iters = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((k_probe32<NW, NI, NJ>), dim3(grid), dim3(64 * NW), 0, 0, src, out, reps);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    const double flop = (double)grid * NW * reps * STEPS * NI * NJ * 2.0 * 2.0 * 32 * 32 * 16;
    printf("%-44s grid %5d: %.3f ms  %.1f TFLOP/s   (%d reads per %d MFMAs 32x32x16 per wave step)\n", name, grid, ms, flop / ms / 1e9, 2 * (NI + NJ), 2 * NI * NJ);
}

template <int NW, int NI, int NJ>
static void run(const char *name, const uint4 *src, float *out, int grid, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_probe<NW, NI, NJ>), dim3(grid), dim3(64 * NW), 0, 0, src, out, reps);
    CK(hipDeviceSynchronize());
    const int iters = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((k_probe<NW, NI, NJ>), dim3(grid), dim3(64 * NW), 0, 0, src, out, reps);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    const double flop = (double)grid * NW * reps * STEPS * NI * NJ * 2.0 * 16 * 16 * 32;
    printf("%-44s grid %5d: %.3f ms  %.1f TFLOP/s   (%d reads per %d MFMAs per wave step)\n", name, grid, ms, flop / ms / 1e9, NI + NJ, NI * NJ);
}

int main(int argc, char **argv)
{
    const bool relu = getenv("RELU") != nullptr;
    std::vector<unsigned short> h(LDS_BYTES / 2);
    srand(1);
    for (auto &v : h) {
        float f = (float)rand() / RAND_MAX * 2.f - 1.f;
        if (relu && f < 0) f = 0.f;
        unsigned u; memcpy(&u, &f, 4);
        v = (unsigned short)(u >> 16);
    }
    uint4 *src; float *out;
    CK(hipMalloc(&src, LDS_BYTES)); CK(hipMalloc(&out, 8192 * 512 * 4));
    CK(hipMemcpy(src, h.data(), LDS_BYTES, hipMemcpyHostToDevice));
    const int reps = 4;
    // same work per launch: 5632 tiles of 256 x 256 (x reps); 8 waves of 64ch x 128pt, or 4 waves of 128 x 128
    run<8, 4, 8>("8 waves x (64 ch x 128 pt), 2 waves/SIMD", src, out, 5632, reps);
    run<4, 8, 8>("4 waves x (128 ch x 128 pt), 1 wave/SIMD", src, out, 5632, reps);
    run<8, 4, 8>("8 waves x (64 ch x 128 pt), 2 waves/SIMD", src, out, 5632, reps);
    run<4, 8, 8>("4 waves x (128 ch x 128 pt), 1 wave/SIMD", src, out, 5632, reps);
    run<4, 4, 8>("4 waves x (64 ch x 128 pt), 1 wave/SIMD", src, out, 5632 * 2, reps);
    run32<8, 2, 4>("8 waves x (64 ch x 128 pt), 32x32x16 MFMA", src, out, 5632, reps);
    run<8, 4, 8>("8 waves x (64 ch x 128 pt), 2 waves/SIMD", src, out, 5632, reps);
    run32<8, 2, 4>("8 waves x (64 ch x 128 pt), 32x32x16 MFMA", src, out, 5632, reps);
    return 0;
}
