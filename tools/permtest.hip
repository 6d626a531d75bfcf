#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(uint32_t a, uint32_t b, uint32_t* out) {
    uint32_t s = threadIdx.x;  // selector byte 0..255 in byte 0; bytes 1..3 fixed selectors 0,4,12
    uint32_t sel = s | (0u << 8) | (4u << 16) | (12u << 24);
    out[threadIdx.x] = __builtin_amdgcn_perm(a, b, sel);
}
int main() {
    uint32_t* d; hipMalloc(&d, 1024);
    uint32_t a = 0x87868584u, b = 0x83020100u;  // S0 bytes: 84 85 86 87 ; S1 bytes: 00 01 02 83
    k<<<1, 256>>>(a, b, d);
    uint32_t h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 256; i++) if (i < 20 || (i & 15) == 0 || i > 250) printf("sel %3d -> %08x\n", i, h[i]);
    // summary: all sel>=13 give ff?
    int ok = 1; for (int i = 13; i < 256; i++) if ((h[i] & 0xff) != 0xff) { ok = 0; printf("sel %d low byte %02x\n", i, h[i] & 0xff); }
    printf("all sel>=13 -> ff: %d\n", ok);
    return 0;
}
