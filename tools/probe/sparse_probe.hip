// Probe: does the memory system reward skipping the quality bytes of a BAM record?
// Streams a buffer laid out like the C3 record stream (277-B records, byte-aligned) two ways:
//   full   : every 16-B chunk (what stage_tile_dma does today)
//   sparse : only the chunks covering [o, o+65) and [o+114, o+128) of each record
// and prints records/s for both.  Standalone: hipcc --offload-arch=gfx950 -O3 -o sparse_probe sparse_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void full_kernel(const uint4 *__restrict__ p, size_t n16, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        v4u v = __builtin_nontemporal_load((const v4u *)(p + i));
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

// 8 lanes per record; lane l<5 -> head chunk l, l=5,6 -> tail chunks, l=7 idle
template <int GRAN>
__global__ __launch_bounds__(256) void sparse_kernel(const unsigned char *__restrict__ base, size_t n_rec, unsigned rec_bytes,
                                                     unsigned head_bytes, unsigned tail_off, unsigned tail_bytes, unsigned *sink)
{
    unsigned acc = 0;
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const unsigned sub = tid & 7;
    for (size_t r = tid >> 3; r < n_rec; r += ((size_t)gridDim.x * blockDim.x) >> 3) {
        const size_t o = r * rec_bytes;
        const size_t h0 = o & ~(size_t)(GRAN - 1), h1 = (o + head_bytes - 1) & ~(size_t)(GRAN - 1);
        const size_t t0 = (o + tail_off) & ~(size_t)(GRAN - 1), t1 = (o + tail_off + tail_bytes - 1) & ~(size_t)(GRAN - 1);
        // enumerate distinct GRAN-byte pieces: h0..h1 then t0..t1 (skipping overlap)
        const unsigned nh = (unsigned)((h1 - h0) / GRAN) + 1;
        size_t ts = t0 <= h1 ? h1 + GRAN : t0;
        const unsigned nt = ts > t1 ? 0 : (unsigned)((t1 - ts) / GRAN) + 1;
        const unsigned per = GRAN / 16;   // 16-B loads per piece
        for (unsigned c = sub; c < (nh + nt) * per; c += 8) {
            const unsigned piece = c / per, within = c % per;
            const size_t a = (piece < nh ? h0 + (size_t)piece * GRAN : ts + (size_t)(piece - nh) * GRAN) + within * 16;
            v4u v = __builtin_nontemporal_load((const v4u *)(base + a));
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char **argv)
{
    size_t n_rec = argc > 1 ? strtoull(argv[1], 0, 10) : 60000000ull;
    unsigned rec = argc > 2 ? atoi(argv[2]) : 277, head = argc > 3 ? atoi(argv[3]) : 65, toff = argc > 4 ? atoi(argv[4]) : 114,
             tlen = argc > 5 ? atoi(argv[5]) : 14;
    size_t bytes = n_rec * rec + 256;
    unsigned char *d; unsigned *sink;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(d, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 8, iters = 10;
    float ms;
    auto time = [&](auto launch, const char *name) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; i++) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters;
        printf("%-12s %8.3f ms  %7.2f G records/s  (%.2f TB/s of whole records)\n", name, ms, n_rec / ms / 1e6, n_rec * (double)rec / ms / 1e9);
    };
    printf("records %zu x %u B, need [0,%u) + [%u,%u)\n", n_rec, rec, head, toff, toff + tlen);
    time([&] { full_kernel<<<grid, 256>>>((const uint4 *)d, bytes / 16, sink); }, "full");
    time([&] { sparse_kernel<16><<<grid * 4, 256>>>(d, n_rec, rec, head, toff, tlen, sink); }, "sparse16");
    time([&] { sparse_kernel<32><<<grid * 4, 256>>>(d, n_rec, rec, head, toff, tlen, sink); }, "sparse32");
    time([&] { sparse_kernel<64><<<grid * 4, 256>>>(d, n_rec, rec, head, toff, tlen, sink); }, "sparse64");
    time([&] { sparse_kernel<128><<<grid * 4, 256>>>(d, n_rec, rec, head, toff, tlen, sink); }, "sparse128");
    return 0;
}
