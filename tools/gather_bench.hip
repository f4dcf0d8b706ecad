// Micro-benchmark: how fast can MI355X gather random 256-byte fp32 rows and reduce them, and with which
// instruction shape?  This pins the ceiling the SpMM kernel is measured against (methodology: never infer a
// platform ceiling from your own slow kernel).  Each wave reduces `per_wave` random rows into one output row.
//
//   K1  lane == column, v_readlane index -> 64-bit address -> global_load_dword             (what spmm v1/v2 do)
//   K2  lane == column, v_readlane pre-scaled byte offset -> buffer_load_dword soffset      (no scalar address math)
//   K3  as K2 but indices / values come in by wave-uniform (scalar) loads, 16 at a time
//   K4  16 lanes x float4 per row, 4 rows per wave instruction, per-group index stream      (what spmm v3 does)
//
// build: hipcc -O3 --offload-arch=gfx950 tools/gather_bench.hip -o gather_bench ; run: ./gather_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e = (x);                                                            \
        if (e != hipSuccess) {                                                         \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), /*stride*/ 0, (int)bytes, 0x00020000);
}

constexpr int U = 16;

template <int UU>
__global__ __launch_bounds__(256) void k1u(const float *__restrict__ X, const int *__restrict__ idx,
                                           const float *__restrict__ val, float *__restrict__ Y, int per_wave)
{
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int *ip = idx + (size_t)w * per_wave;
    const float *vp = val + (size_t)w * per_wave;
    float acc = 0.f;
    for (int base = 0; base < per_wave; base += 64) {
        const int my_i = ip[base + lane];
        const float my_v = vp[base + lane];
        for (int i = 0; i < 64; i += UU) {
            float x[UU];
#pragma unroll
            for (int u = 0; u < UU; ++u) x[u] = X[(size_t)__builtin_amdgcn_readlane(my_i, i + u) * 64 + lane];
#pragma unroll
            for (int u = 0; u < UU; ++u)
                acc = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), i + u)), x[u], acc);
        }
    }
    Y[(size_t)w * 64 + lane] = acc;
}

__global__ __launch_bounds__(1024) void k1_wg1024(const float *__restrict__ X, const int *__restrict__ idx,
                                                   const float *__restrict__ val, float *__restrict__ Y, int per_wave)
{
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 16 + (threadIdx.x >> 6);
    const int *ip = idx + (size_t)w * per_wave;
    const float *vp = val + (size_t)w * per_wave;
    float acc = 0.f;
    for (int base = 0; base < per_wave; base += 64) {
        const int my_i = ip[base + lane];
        const float my_v = vp[base + lane];
        for (int i = 0; i < 64; i += U) {
            float x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = X[(size_t)__builtin_amdgcn_readlane(my_i, i + u) * 64 + lane];
#pragma unroll
            for (int u = 0; u < U; ++u)
                acc = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), i + u)), x[u], acc);
        }
    }
    Y[(size_t)w * 64 + lane] = acc;
}

// K1 behind a dependent task-descriptor load (what a task table costs)
__global__ __launch_bounds__(256) void k1_desc(const float *__restrict__ X, const int *__restrict__ idx,
                                               const float *__restrict__ val, float *__restrict__ Y, int per_wave,
                                               const int *__restrict__ desc)
{
    const int lane = threadIdx.x & 63;
    const int w0 = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int w = desc[w0];
    const int *ip = idx + (size_t)w * per_wave;
    const float *vp = val + (size_t)w * per_wave;
    float acc = 0.f;
    for (int base = 0; base < per_wave; base += 64) {
        const int my_i = ip[base + lane];
        const float my_v = vp[base + lane];
        for (int i = 0; i < 64; i += U) {
            float x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = X[(size_t)__builtin_amdgcn_readlane(my_i, i + u) * 64 + lane];
#pragma unroll
            for (int u = 0; u < U; ++u)
                acc = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), i + u)), x[u], acc);
        }
    }
    Y[(size_t)w * 64 + lane] = acc;
}

__global__ __launch_bounds__(256) void k1(const float *__restrict__ X, const int *__restrict__ idx,
                                          const float *__restrict__ val, float *__restrict__ Y, int per_wave)
{
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int *ip = idx + (size_t)w * per_wave;
    const float *vp = val + (size_t)w * per_wave;
    float acc = 0.f;
    for (int base = 0; base < per_wave; base += 64) {
        const int my_i = ip[base + lane];
        const float my_v = vp[base + lane];
        for (int i = 0; i < 64; i += U) {
            float x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = X[(size_t)__builtin_amdgcn_readlane(my_i, i + u) * 64 + lane];
#pragma unroll
            for (int u = 0; u < U; ++u)
                acc = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), i + u)), x[u], acc);
        }
    }
    Y[(size_t)w * 64 + lane] = acc;
}

__global__ __launch_bounds__(256) void k2(const float *__restrict__ X, uint32_t xbytes, const int *__restrict__ off,
                                          const float *__restrict__ val, float *__restrict__ Y, int per_wave)
{
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int *ip = off + (size_t)w * per_wave;
    const float *vp = val + (size_t)w * per_wave;
    const rsrc_t rsrc = make_rsrc(X, xbytes);
    float acc = 0.f;
    for (int base = 0; base < per_wave; base += 64) {
        const int my_o = ip[base + lane];  // byte offset of the row
        const float my_v = vp[base + lane];
        for (int i = 0; i < 64; i += U) {
            float x[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                x[u] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4, __builtin_amdgcn_readlane(my_o, i + u), 0));
#pragma unroll
            for (int u = 0; u < U; ++u)
                acc = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), i + u)), x[u], acc);
        }
    }
    Y[(size_t)w * 64 + lane] = acc;
}

// scalar metadata: uniform-index loads (the compiler should pick s_load_dwordx*), consumed straight from SGPRs
__global__ __launch_bounds__(256) void k3(const float *__restrict__ X, uint32_t xbytes, const int *__restrict__ off,
                                          const float *__restrict__ val, float *__restrict__ Y, int per_wave)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int *ip = off + (size_t)w * per_wave;
    const float *vp = val + (size_t)w * per_wave;
    const rsrc_t rsrc = make_rsrc(X, xbytes);
    float acc = 0.f;
    for (int i = 0; i < per_wave; i += U) {
        int o[U];
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            o[u] = ip[i + u];
            v[u] = vp[i + u];
        }
        float x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4, o[u], 0));
#pragma unroll
        for (int u = 0; u < U; ++u) acc = fmaf(v[u], x[u], acc);
    }
    Y[(size_t)w * 64 + lane] = acc;
}

// 16 lanes x float4, four independent index streams per wave (per_wave/4 rows each), UQ gathers in flight per group
template <int UQ>
__global__ __launch_bounds__(256) void k4(const float *__restrict__ X, const int *__restrict__ idx,
                                          const float *__restrict__ val, float *__restrict__ Y, int per_wave)
{
    const int lane = threadIdx.x & 63, q = lane & 15, g = lane >> 4;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int per_group = per_wave / 4;
    const int *ip = idx + (size_t)w * per_wave + g * per_group;
    const float *vp = val + (size_t)w * per_wave + g * per_group;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int i = 0; i < per_group; i += UQ) {
        float4 x[UQ];
        float v[UQ];
#pragma unroll
        for (int u = 0; u < UQ; ++u) {
            const int c = ip[i + u];
            v[u] = vp[i + u];
            x[u] = *reinterpret_cast<const float4 *>(X + (size_t)c * 64 + q * 4);
        }
#pragma unroll
        for (int u = 0; u < UQ; ++u) {
            acc.x = fmaf(v[u], x[u].x, acc.x);
            acc.y = fmaf(v[u], x[u].y, acc.y);
            acc.z = fmaf(v[u], x[u].z, acc.z);
            acc.w = fmaf(v[u], x[u].w, acc.w);
        }
    }
    // fold the four group sums (order irrelevant for the benchmark)
    acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16); acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
    acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
    if (g == 0) *reinterpret_cast<float4 *>(Y + (size_t)w * 64 + q * 4) = acc;
}

int main(int argc, char **argv)
{
    const int per_wave = 64;
    struct Cfg { size_t rows; size_t waves; const char *name; };
    Cfg cfgs[] = {{15593, 6544, "4 MB table (Epinion2-sized, cache-resident), 418k gathers = one Epinion2 layer"},
                  {15593, 6540 * 4, "4 MB table (Epinion2-sized, cache-resident), 1.67M gathers"},
                  {15593, 6540 * 16, "4 MB table (Epinion2-sized, cache-resident), 6.7M gathers"},
                  {1u << 23, 1u << 20, "2 GB table (HBM-resident), 67M gathers"}};
    for (const Cfg &c : cfgs) {
        const size_t n_idx = c.waves * per_wave;
        float *X, *Y, *val;
        int *idx, *off;
        CK(hipMalloc(&X, c.rows * 256));
        CK(hipMalloc(&Y, c.waves * 256));
        CK(hipMalloc(&val, n_idx * 4));
        CK(hipMalloc(&idx, n_idx * 4));
        CK(hipMalloc(&off, n_idx * 4));
        std::vector<int> h(n_idx), ho(n_idx);
        std::vector<float> hv(n_idx, 0.5f), hx(c.rows * 64, 1.0f);
        uint64_t s = 88172645463325252ull;
        for (size_t i = 0; i < n_idx; ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            h[i] = (int)(s % c.rows);
            ho[i] = h[i] * 256;
        }
        CK(hipMemcpy(idx, h.data(), n_idx * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(off, ho.data(), n_idx * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(val, hv.data(), n_idx * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(X, hx.data(), c.rows * 256, hipMemcpyHostToDevice));
        const dim3 grid((unsigned)(c.waves / 4)), block(256);
        const uint32_t xbytes = (uint32_t)(c.rows * 256 > 0xffffffffull ? 0xffffffffull : c.rows * 256);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        printf("== %s\n", c.name);
        int *desc;
        CK(hipMalloc(&desc, c.waves * 4));
        { std::vector<int> hd(c.waves); for (size_t i = 0; i < c.waves; ++i) hd[i] = (int)i; CK(hipMemcpy(desc, hd.data(), c.waves * 4, hipMemcpyHostToDevice)); }
        for (int k = 1; k <= 11; ++k) {
            float best = 1e30f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                switch (k) {
                    case 1: hipLaunchKernelGGL(k1, grid, block, 0, 0, X, idx, val, Y, per_wave); break;
                    case 2: hipLaunchKernelGGL(k2, grid, block, 0, 0, X, xbytes, off, val, Y, per_wave); break;
                    case 3: hipLaunchKernelGGL(k3, grid, block, 0, 0, X, xbytes, off, val, Y, per_wave); break;
                    case 4: hipLaunchKernelGGL(k4<4>, grid, block, 0, 0, X, idx, val, Y, per_wave); break;
                    case 5: hipLaunchKernelGGL(k4<8>, grid, block, 0, 0, X, idx, val, Y, per_wave); break;
                    case 6: hipLaunchKernelGGL(k4<16>, grid, block, 0, 0, X, idx, val, Y, per_wave); break;
                    case 7: hipLaunchKernelGGL(k1u<32>, grid, block, 0, 0, X, idx, val, Y, per_wave); break;
                    case 8: hipLaunchKernelGGL(k1u<64>, grid, block, 0, 0, X, idx, val, Y, per_wave); break;
                    case 9: hipLaunchKernelGGL(k1u<8>, grid, block, 0, 0, X, idx, val, Y, per_wave); break;
                    case 10: hipLaunchKernelGGL(k1_wg1024, dim3((unsigned)(c.waves / 16)), dim3(1024), 0, 0, X, idx, val, Y, per_wave); break;
                    case 11: hipLaunchKernelGGL(k1_desc, grid, block, 0, 0, X, idx, val, Y, per_wave, desc); break;
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            std::vector<float> hy(64);
            CK(hipMemcpy(hy.data(), Y, 256, hipMemcpyDeviceToHost));
            const double gb = (double)n_idx * 264.0 / 1e9;
            printf("  K%d%s  %9.1f us   %7.0f GB/s (264 B/gather)   %6.2f G gathers/s   check %.1f\n", k,
                   k == 4 ? "(x4,4 in flight)" : k == 5 ? "(x4,8)" : k == 6 ? "(x4,16)" : k == 7 ? "(K1, 32 in flight)" : k == 8 ? "(K1, 64 in flight)" : k == 9 ? "(K1, 8 in flight)" : k == 10 ? "(K1, 1024-thread blocks)" : k == 11 ? "(K1 behind a descriptor load)" : "", best * 1e3, gb / (best * 1e-3),
                   n_idx / (best * 1e-3) / 1e9, hy[0]);
        }
        CK(hipFree(desc));
        CK(hipFree(X)); CK(hipFree(Y)); CK(hipFree(val)); CK(hipFree(idx)); CK(hipFree(off));
    }
    return 0;
}
