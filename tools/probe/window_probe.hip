// Probe for the "2-bit reference image" question: what do the two window gathers of a SHUFFLED 150-bp read
// cost against a 4-bit image (what the tiled kernels read today: 16 B per end) and against a 2-bit image (8 B per end,
// with and without a second gather into a 1-bit "other letter" mask)?  Read starts are a hash of the read's index over a
// 3.1 Gb genome (no two neighbours share a line), the right window starts L - 32 positions behind the left one.
// Prints reads/s and the bytes per read that the time corresponds to at the measured streaming rate of the same launch.
// Standalone: hipcc --offload-arch=gfx950 -O3 -o window_probe window_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// MODE 0: 4 bit/base, 16 B per end.  1: 2 bit/base, 8 B per end.  2: 2 bit/base + 1 bit/base mask (4 B per end).
// 3: nothing but the record stream (277 B per read, sequential) -- the part every mode shares, for the scale
template <int MODE>
__global__ __launch_bounds__(256) void probe(const unsigned char *__restrict__ img, const unsigned char *__restrict__ mask,
                                             const uint4 *__restrict__ recs, uint64_t n_reads, uint64_t genome, unsigned L, unsigned *sink)
{
    unsigned acc = 0;
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t s = mix(r) % (genome - 1024);
        const uint64_t w[2] = {s, s + L - 32};
#pragma unroll
        for (int e = 0; e < 2; e++) {
            if (MODE == 0) { v4u v = *(const v4u *)(img + ((w[e] >> 3) << 2)); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
            if (MODE == 1 || MODE == 2) { v2u v = *(const v2u *)(img + ((w[e] >> 4) << 2)); acc ^= v.x ^ v.y; }
            if (MODE == 2) acc ^= *(const unsigned *)(mask + ((w[e] >> 5) << 2));
        }
    }
    if (MODE == 3) {
        const uint64_t n16 = n_reads * 277 / 16;
        for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
            v4u v = __builtin_nontemporal_load((const v4u *)(recs + i));
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char **argv)
{
    const uint64_t n_reads = argc > 1 ? strtoull(argv[1], 0, 10) : 14285714ull, genome = 3100000000ull;
    const unsigned L = argc > 2 ? atoi(argv[2]) : 150;
    unsigned char *img, *mask; uint4 *recs; unsigned *sink;
    CK(hipMalloc(&img, genome / 2 + 4096)); CK(hipMalloc(&mask, genome / 8 + 4096)); CK(hipMalloc(&recs, n_reads * 277 + 4096)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(img, 1, genome / 2 + 4096)); CK(hipMemset(mask, 0, genome / 8 + 4096)); CK(hipMemset(recs, 2, n_reads * 277 + 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 8, iters = 10;
    const char *name[4] = {"4 bit/base, 16 B per end            ", "2 bit/base,  8 B per end            ", "2 bit/base + 1 bit/base mask        ",
                           "record stream alone (277 B per read)"};
    float ms[4];
    for (int m = 0; m < 4; m++) {
        for (int it = 0; it < iters + 2; it++) {
            if (it == 2) CK(hipEventRecord(e0));
            if (m == 0) probe<0><<<grid, 256>>>(img, mask, recs, n_reads, genome, L, sink);
            if (m == 1) probe<1><<<grid, 256>>>(img, mask, recs, n_reads, genome, L, sink);
            if (m == 2) probe<2><<<grid, 256>>>(img, mask, recs, n_reads, genome, L, sink);
            if (m == 3) probe<3><<<grid, 256>>>(img, mask, recs, n_reads, genome, L, sink);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[m], e0, e1));
        ms[m] /= iters;
    }
    const double stream_Bps = n_reads * 277.0 / (ms[3] * 1e-3);
    for (int m = 0; m < 4; m++)
        printf("%s : %.3f ms per %llu reads = %.2f G reads/s; at the stream's %.2f TB/s that time moves %.0f B per read\n", name[m], ms[m],
               (unsigned long long)n_reads, n_reads / ms[m] * 1e-6, stream_Bps * 1e-12, ms[m] * 1e-3 * stream_Bps / n_reads);
    return 0;
}
