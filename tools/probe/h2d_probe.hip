// Probe: what host-to-device rate do page-locked 32 MiB chunks reach (a) as hipMemcpyAsync on 1 / 2 / 4 streams, (b) pulled by a
// copy KERNEL that reads the page-locked host memory itself, (c) both at once?  (The compressed feed of bin/pss-bam waits
// for its copies for 0.2 of its 0.38 s: device_feed.c "waiting for copies".)
// Standalone: hipcc --offload-arch=gfx950 -O3 -o h2d_probe h2d_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <chrono>
#include <sys/mman.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void pull(const v4u *__restrict__ src, v4u *__restrict__ dst, size_t n16)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = __builtin_nontemporal_load(src + i);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const size_t chunk = 32ull << 20, n_chunks = argc > 1 ? strtoull(argv[1], 0, 10) : 128;
    const size_t ring = 16;
    unsigned char *h, *d;
    CK(hipHostMalloc((void **)&h, ring * chunk, hipHostMallocDefault));
    memset(h, 3, ring * chunk);
    CK(hipMalloc(&d, ring * chunk));
    hipStream_t st[5];
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int n_st : {1, 2, 4}) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (size_t k = 0; k < n_chunks; k++) CK(hipMemcpyAsync(d + (k % ring) * chunk, h + (k % ring) * chunk, chunk, hipMemcpyHostToDevice, st[k % n_st]));
        CK(hipDeviceSynchronize());
        const double t = now() - t0;
        printf("hipMemcpyAsync, %d stream(s)            : %.1f GB/s\n", n_st, n_chunks * chunk / t * 1e-9);
    }
    for (int wgs : {32, 128, 512}) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (size_t k = 0; k < n_chunks; k++) pull<<<wgs, 256, 0, st[4]>>>((const v4u *)(h + (k % ring) * chunk), (v4u *)(d + (k % ring) * chunk), chunk / 16);
        CK(hipDeviceSynchronize());
        const double t = now() - t0;
        printf("copy kernel, %3d workgroups of 256        : %.1f GB/s\n", wgs, n_chunks * chunk / t * 1e-9);
    }
    {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (size_t k = 0; k < n_chunks; k++) {
            if (k & 1) pull<<<128, 256, 0, st[4]>>>((const v4u *)(h + (k % ring) * chunk), (v4u *)(d + (k % ring) * chunk), chunk / 16);
            else CK(hipMemcpyAsync(d + (k % ring) * chunk, h + (k % ring) * chunk, chunk, hipMemcpyHostToDevice, st[0]));
        }
        CK(hipDeviceSynchronize());
        const double t = now() - t0;
        printf("alternating: hipMemcpyAsync / copy kernel : %.1f GB/s\n", n_chunks * chunk / t * 1e-9);
    }
    // the same copies from memory that was allocated by the host first and page-locked afterwards (what device_feed.c's staging
    // slots are: the loaders start filling them before the HIP runtime is up): 4 KiB pages / transparent huge pages
    for (int huge = 0; huge < 2; huge++) {
        unsigned char *m = nullptr;
        if (posix_memalign((void **)&m, huge ? (2u << 20) : 4096, ring * chunk) != 0) return 1;
        if (huge) madvise(m, ring * chunk, MADV_HUGEPAGE);
        memset(m, 5, ring * chunk);
        const double tr = now();
        CK(hipHostRegister(m, ring * chunk, hipHostRegisterDefault));
        const double t_reg = now() - tr;
        for (int n_st : {1, 2}) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            for (size_t k = 0; k < n_chunks; k++) CK(hipMemcpyAsync(d + (k % ring) * chunk, m + (k % ring) * chunk, chunk, hipMemcpyHostToDevice, st[k % n_st]));
            CK(hipDeviceSynchronize());
            const double t = now() - t0;
            printf("hipHostRegister'd (%s, registering 512 MiB took %.3f s), %d stream(s): %.1f GB/s\n", huge ? "2 MiB-aligned + MADV_HUGEPAGE" : "4 KiB pages", t_reg, n_st,
                   n_chunks * chunk / t * 1e-9);
        }
        CK(hipHostUnregister(m));
        free(m);
    }
    return 0;
}
