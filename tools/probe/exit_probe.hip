// tools/probe/exit_probe.hip -- how long does the kernel take to tear a HIP process down?  Allocates `vram_gb` of device
// memory in 4 GiB pieces (touched by a memset), page-locks `pin_mb` of host memory, maps + touches `host_gb` of anonymous
// memory, then leaves in the way argv[4] says: "exit" (_exit at once), "free" (hipFree / unregister everything first),
// "reset" (hipDeviceReset first).  Prints the seconds it spent before leaving; the caller times the whole process.
//   hipcc --offload-arch=gfx950 -O2 -o exit_probe exit_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <pthread.h>
#include <unistd.h>
#include <vector>
static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
int main(int argc, char **argv) {
    const double t0 = now();
    const double vram_gb = argc > 1 ? atof(argv[1]) : 0, pin_mb = argc > 2 ? atof(argv[2]) : 0, host_gb = argc > 3 ? atof(argv[3]) : 0;
    const char *how = argc > 4 ? argv[4] : "exit";
    const int n_streams = argc > 5 ? atoi(argv[5]) : 0, n_threads = argc > 6 ? atoi(argv[6]) : 0;
    hipFree(nullptr);
    const double t_init = now() - t0;
    std::vector<void *> bufs;
    for (double left = vram_gb; left > 0; left -= 4) {
        void *p = nullptr;
        const size_t n = (size_t)((left < 4 ? left : 4) * (1ull << 30));
        if (hipMalloc(&p, n) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
        hipMemsetAsync(p, 1, n, nullptr);
        bufs.push_back(p);
    }
    void *pin = nullptr;
    const size_t pin_n = (size_t)(pin_mb * (1 << 20));
    if (pin_n) { posix_memalign(&pin, 4096, pin_n); memset(pin, 1, pin_n); hipHostRegister(pin, pin_n, hipHostRegisterDefault); }
    char *host = nullptr;
    const size_t host_n = (size_t)(host_gb * (1ull << 30));
    if (host_n) { host = (char *)malloc(host_n); for (size_t i = 0; i < host_n; i += 4096) host[i] = 1; }
    std::vector<hipStream_t> streams((size_t)n_streams);
    void *small = nullptr;
    hipMalloc(&small, 1 << 20);
    for (int i = 0; i < n_streams; i++) {   // (a stream gets its hardware queue at first use)
        hipStreamCreateWithFlags(&streams[(size_t)i], hipStreamNonBlocking);
        hipMemsetAsync(small, i, 1 << 20, streams[(size_t)i]);
    }
    for (int i = 0; i < n_threads; i++) { pthread_t th; pthread_create(&th, nullptr, [](void *) -> void * { for (;;) pause(); return nullptr; }, nullptr); }
    hipDeviceSynchronize();
    const double t_setup = now() - t0;
    if (!strcmp(how, "free")) {
        for (void *p : bufs) hipFree(p);
        if (pin) hipHostUnregister(pin);
    } else if (!strcmp(how, "reset")) hipDeviceReset();
    fprintf(stderr, "init %.3f setup %.3f pre-exit work %.3f s\n", t_init, t_setup, now() - t0 - t_setup);
    _exit(0);
}
