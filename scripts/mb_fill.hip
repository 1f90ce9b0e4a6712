// build: hipcc --offload-arch=gfx950 -O3 -o fill scripts/mb_fill.hip ; run on the GPU box: ./fill
// microbenchmark: how fast can every CU of the chip bring the same 128 KB of a 176 KB matrix into its LDS?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int L = 186, NODES = 86;            // 86 columns of 186 doubles = 127 968 B
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

template <int MODE>
__global__ __launch_bounds__(1024) void k(const double *src, double *out, unsigned long long *stamps, int reps)
{
    extern __shared__ __align__(16) double sh[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    double acc = 0.0;
    unsigned long long t0 = 0, t1 = 0;
    for (int rep = 0; rep < reps; ++rep) {
        __syncthreads();
        if (rep == reps - 1) t0 = wall_clock64();
        const int rot = MODE >= 10 ? (blockIdx.x * 7) % NODES : 0;
        if (MODE % 10 == 0) {                 // glds, a wave per column, 16-byte pieces
            for (int s0 = wv; s0 < NODES; s0 += 16) {
                int slot = s0 + rot; if (slot >= NODES) slot -= NODES;
                const double *g = src + (size_t)L * slot;
                char *d = (char *)(sh + (size_t)slot * L);
                for (int g0 = 0; g0 < L / 2; g0 += 64)
                    if (g0 + lane < L / 2)
                        __builtin_amdgcn_global_load_lds((glb_void *)(g + 2 * (g0 + lane)), (lds_void *)(d + (size_t)g0 * 16), 16, 0, 0);
            }
        } else if (MODE % 10 == 1) {          // registers: dwordx4 loads of the linear image, ds_write_b128
            const int tot = NODES * L / 2;    // 16-byte pieces
            const double2 *g2 = (const double2 *)src;
            double2 *s2 = (double2 *)sh;
            double2 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { int i = tid + 1024 * u + rot * (L / 2); if (i >= tot) i -= tot; x[u] = (tid + 1024 * u) < tot ? g2[i] : double2{0, 0}; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { int i = tid + 1024 * u + rot * (L / 2); if (i >= tot) i -= tot; if ((tid + 1024 * u) < tot) s2[i] = x[u]; }
        } else if (MODE % 10 == 2) {          // registers, 8-byte loads (the product's streaming width), no LDS: sum only
            const int tot = NODES * L;
            double x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { int i = tid + 1024 * u; x[u] = i < tot ? src[i] : 0.0; }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += x[u];
        } else if (MODE % 10 == 3) {          // the product's dual-half pattern: thread (pl, l), 32 nodes strided by L, two batches of 16
            const int pl = tid >> 8, l = tid & 255;
            for (int b = 0; b < 2; ++b) {
                double x[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { const int n = pl * 32 + b * 16 + u; x[u] = (l < L && n < 118) ? src[l + (size_t)L * n] : 0.0; }
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += x[u];
            }
        }
        __syncthreads();
        if (rep == reps - 1) t1 = wall_clock64();
        if (MODE % 10 <= 1) acc += sh[(tid * 13 + rep) % (NODES * L)];
    }
    out[blockIdx.x * 1024 + tid] = acc;
    if (tid == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
}

template <int MODE>
int run(const char *name, const double *src, double *out, unsigned long long *st, int blocks)
{
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    std::vector<unsigned long long> h(2 * blocks);
    for (int reps : {1, 4}) {
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 130 * 1024, 0, src, out, st, reps);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 130 * 1024, 0, src, out, st, reps);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        double mx = 0, av = 0;
        for (int b = 0; b < blocks; ++b) { const double d = (h[2 * b + 1] - h[2 * b]) / 100.0; mx = d > mx ? d : mx; av += d / blocks; }
        printf("%-44s blocks %3d reps %d: last pass %.2f us mean, %.2f us max\n", name, blocks, reps, av, mx);
    }
    return 0;
}

int main()
{
    double *src, *out; unsigned long long *st;
    CK(hipMalloc(&src, 256 * 256 * 8)); CK(hipMalloc(&out, 256 * 1024 * 8)); CK(hipMalloc(&st, 2 * 256 * 8));
    CK(hipMemset(src, 0, 256 * 256 * 8));
    for (int blocks : {168, 21, 1}) {
        if (run<0>("glds dwordx4, wave per column", src, out, st, blocks)) return 1;
        if (run<10>("glds dwordx4, rotated start per block", src, out, st, blocks)) return 1;
        if (run<1>("registers dwordx4 + ds_write_b128", src, out, st, blocks)) return 1;
        if (run<11>("registers dwordx4 + ds_write, rotated", src, out, st, blocks)) return 1;
        if (run<2>("registers dwordx2, 16 in flight, no LDS", src, out, st, blocks)) return 1;
        if (run<3>("product pattern (pl, l): 2 x 16 strided rows", src, out, st, blocks)) return 1;
    }
    return 0;
}
