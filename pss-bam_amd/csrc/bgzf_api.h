// pss-bam_amd/csrc/bgzf_api.h -- C ABI of the device-side BGZF inflate (included by engine.hip).
#pragma once

#include "inflate_kernels.h"
#include "inflate_wave.h"

// walks BGZF block headers (SAM spec 4.1) over whole blocks; returns the count, or < 0
extern "C" int64_t pssbam_bgzf_scan(const void *bytes, uint64_t nbytes, pssbam_bgzf_block *blocks, uint64_t max_blocks,
                                    uint64_t *consumed, uint64_t *inflated_bytes) {
    const uint8_t *p = (const uint8_t *)bytes;
    uint64_t o = 0, n = 0, uoff = 0;
    while (o + 18 <= nbytes && (!blocks || n < max_blocks)) {
        if (p[o] != 0x1f || p[o + 1] != 0x8b || p[o + 2] != 8 || !(p[o + 3] & 4)) { fail(PSSBAM_EFORMAT, "not a BGZF block at offset %llu", (unsigned long long)o); return PSSBAM_EFORMAT; }
        const uint32_t xlen = p[o + 10] | ((uint32_t)p[o + 11] << 8);
        if (o + 12 + xlen > nbytes) break;
        uint32_t bsize = 0;
        bool found = false;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t *sf = p + o + 12 + x;
            const uint32_t slen = sf[2] | ((uint32_t)sf[3] << 8);
            if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && x + 6 <= xlen) { bsize = (sf[4] | ((uint32_t)sf[5] << 8)) + 1u; found = true; break; }
            x += 4 + slen;
        }
        if (!found || bsize < 12u + xlen + 8u) { fail(PSSBAM_EFORMAT, "BGZF block at offset %llu has no usable BC field", (unsigned long long)o); return PSSBAM_EFORMAT; }
        if (o + bsize > nbytes) break;   // partial block: the caller supplies more bytes
        uint32_t crc, isize;
        memcpy(&crc, p + o + bsize - 8, 4);
        memcpy(&isize, p + o + bsize - 4, 4);
        if (isize > 65536u) { fail(PSSBAM_EFORMAT, "BGZF ISIZE %u exceeds 64 KiB", isize); return PSSBAM_EFORMAT; }
        if (blocks) {
            blocks[n].in_off = o + 12 + xlen;
            blocks[n].in_len = bsize - 12u - xlen - 8u;
            blocks[n].isize = isize;
            blocks[n].out_off = uoff;
            blocks[n].crc = crc;
            blocks[n].status = 0;
        }
        uoff += isize;
        n++;
        o += bsize;
    }
    if (consumed) *consumed = o;
    if (inflated_bytes) *inflated_bytes = uoff;
    return (int64_t)n;
}

namespace {
uint32_t host_gf2_mul(uint32_t a, uint32_t b) {
    uint32_t r = 0;
    for (int i = 0; i < 32; i++) {
        if (b & 0x80000000u) r ^= a;
        b <<= 1;
        a = (a >> 1) ^ ((a & 1u) ? 0xEDB88320u : 0u);
    }
    return r;
}
uint32_t *g_xpow_dev[64] = {nullptr};   // per device: x^(8*1024*k) mod P, k = 0..63
int ensure_xpow(int dev, uint32_t **out) {
    if (dev < 0 || dev >= 64) return fail(PSSBAM_EINVAL, "device %d out of range", dev);
    if (!g_xpow_dev[dev]) {
        uint32_t x1k = 0x80000000u;   // x^0
        for (int i = 0; i < 8 * 1024; i++) x1k = (x1k >> 1) ^ ((x1k & 1u) ? 0xEDB88320u : 0u);
        uint32_t h[64];
        h[0] = 0x80000000u;
        for (int k = 1; k < 64; k++) h[k] = host_gf2_mul(h[k - 1], x1k);
        HIP_TRY(hipMalloc(&g_xpow_dev[dev], sizeof h));
        HIP_TRY(hipMemcpy(g_xpow_dev[dev], h, sizeof h, hipMemcpyHostToDevice));
    }
    *out = g_xpow_dev[dev];
    return PSSBAM_OK;
}
// per device, for bgzf_crc_lines_kernel: x^(8*16*k) mod P (k = 0..63) and the four byte tables of "the register moved on by
// 1024 zero bytes" (Z_j[v] = (v << 8j) * x^8192 mod P)
uint32_t *g_crc2_dev[64] = {nullptr};   // [64 + 4 * 256]
int ensure_crc2_tabs(int dev, uint32_t **xpow16, uint32_t **ztab) {
    if (dev < 0 || dev >= 64) return fail(PSSBAM_EINVAL, "device %d out of range", dev);
    if (!g_crc2_dev[dev]) {
        std::vector<uint32_t> h(64 + 4 * 256);
        uint32_t x16 = 0x80000000u;   // x^0
        for (int i = 0; i < 8 * 16; i++) x16 = (x16 >> 1) ^ ((x16 & 1u) ? 0xEDB88320u : 0u);
        h[0] = 0x80000000u;
        for (int k = 1; k < 64; k++) h[k] = host_gf2_mul(h[k - 1], x16);
        uint32_t t0[256];   // the byte table: the register after one more byte
        for (uint32_t v = 0; v < 256; v++) {
            uint32_t c = v;
            for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1u) ? 0xEDB88320u : 0u);
            t0[v] = c;
        }
        for (uint32_t j = 0; j < 4; j++)
            for (uint32_t v = 0; v < 256; v++) {
                uint32_t c = v << (8 * j);
                for (int n = 0; n < 1024; n++) c = t0[c & 0xFFu] ^ (c >> 8);
                h[64 + j * 256 + v] = c;
            }
        HIP_TRY(hipMalloc(&g_crc2_dev[dev], h.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(g_crc2_dev[dev], h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    *xpow16 = g_crc2_dev[dev];
    *ztab = g_crc2_dev[dev] + 64;
    return PSSBAM_OK;
}
}  // namespace

static_assert(sizeof(pssbam_bgzf_block) == sizeof(pssbam::BgzfBlock), "public and device block descriptors must match");

constexpr int INFLATE_LOOP_DEFAULT = 1;   // what the feed and the host convenience call use (their buffers have the slack)

// loop: 0 = the in-place data loop, 1 = one wait per step (csrc/inflate_kernels.h); $PSSBAM_INFLATE_LOOP overrides
// (except for -2 = the public entry point, which is always 0).  The second loop requests up to 24 bytes past a block's end in d_out and re-reads the
// last 16 bytes of d_comp: callers that ask for it own buffers with that slack.
// out_bytes: what the batch inflates to (0 = unknown) -- picks the literal budget of the second loop.
// what the wave-per-block path needs beside the block table (csrc/inflate_wave.h): the sequence arena, every block's
// first entry in it (prefix sums of seq_cap_of(isize), worked out on the host) and a count per block
struct WaveBufs { uint32_t *seq_arena; const uint64_t *seq_off; uint32_t *seq_count; };
// OFF unless asked for (PSSBAM_INFLATE_WAVE=1; 2 = without the second chance for blocks it hands back): measured slower
// than the lane-per-block kernel on every stream so far (profiles/r03_inflate_wave.txt) -- kept as the experiment it is,
// behind pssbam_bgzf_inflate_host only, bit-exact against zlib in tests/test_gpu_inflate.py
static bool wave_inflate_wanted() {
    const char *v = getenv("PSSBAM_INFLATE_WAVE");
    return v ? atoi(v) != 0 : false;
}

// ISIZE + CRC-32 of every inflated block against its trailer (bgzf_crc_kernel: a block per wave)
static int launch_crc(hipStream_t st, pssbam_bgzf_block *d_blocks, uint32_t n_blocks, void *d_out, bool beside_inflate = false) {
    int dev = 0, n_cu = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    uint32_t *xpow = nullptr;
    int rc = ensure_xpow(dev, &xpow);
    if (rc) return rc;
    static bool crc_attr_set[64] = {false};
    if (!crc_attr_set[dev & 63]) {
        HIP_TRY(hipFuncSetAttribute((const void *)pssbam::bgzf_crc_kernel<pssbam::CRC_REP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pssbam::CRC_LDS_BYTES));
        crc_attr_set[dev & 63] = true;
    }
    static const int crc_form = getenv("PSSBAM_CRC_FORM") ? atoi(getenv("PSSBAM_CRC_FORM")) : 1;   // 1: whole lines per request (bgzf_crc_lines_kernel), 0: a 1 KiB chunk per lane
    if (crc_form == 1 && !beside_inflate) {
        uint32_t *xpow16 = nullptr, *ztab = nullptr;
        rc = ensure_crc2_tabs(dev, &xpow16, &ztab);
        if (rc) return rc;
        static bool crc2_attr_set[64] = {false};
        if (!crc2_attr_set[dev & 63]) {
            HIP_TRY(hipFuncSetAttribute((const void *)pssbam::bgzf_crc_lines_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pssbam::CRC2_LDS_BYTES));
            crc2_attr_set[dev & 63] = true;
        }
        const uint32_t cgrid = std::min<uint32_t>((n_blocks + 15) / 16, (uint32_t)n_cu);
        hipLaunchKernelGGL(pssbam::bgzf_crc_lines_kernel, dim3(cgrid), dim3(1024), pssbam::CRC2_LDS_BYTES, st, (const uint8_t *)d_out, (pssbam::BgzfBlock *)d_blocks, n_blocks,
                           (const uint32_t *)xpow16, (const uint32_t *)ztab);
        HIP_TRY(hipGetLastError());
        return PSSBAM_OK;
    }
    if (beside_inflate) {   // four waves and 16 KiB of tables per workgroup: fits a CU that runs the inflate kernel
        const uint32_t cgrid = std::min<uint32_t>((n_blocks + 3) / 4, (uint32_t)n_cu);
        hipLaunchKernelGGL(pssbam::bgzf_crc_kernel<4u>, dim3(cgrid), dim3(256), 4u * 256u * 4u * 4u, st, (const uint8_t *)d_out, (pssbam::BgzfBlock *)d_blocks, n_blocks, xpow);
    } else {
        // one workgroup per CU, a block per wave (PSSBAM_CRC_WAVES: experiments with fewer lines in flight)
        static const uint32_t crc_waves = getenv("PSSBAM_CRC_WAVES") ? (uint32_t)std::min(16, std::max(1, atoi(getenv("PSSBAM_CRC_WAVES")))) : 16u;
        const uint32_t cgrid = std::min<uint32_t>((n_blocks + crc_waves - 1) / crc_waves, (uint32_t)n_cu);
        hipLaunchKernelGGL(pssbam::bgzf_crc_kernel<pssbam::CRC_REP>, dim3(cgrid), dim3(64 * crc_waves), pssbam::CRC_LDS_BYTES, st, (const uint8_t *)d_out, (pssbam::BgzfBlock *)d_blocks, n_blocks, xpow);
    }
    HIP_TRY(hipGetLastError());
    return PSSBAM_OK;
}

static int launch_inflate(hipStream_t st, const void *d_comp, uint64_t comp_bytes, pssbam_bgzf_block *d_blocks, uint32_t n_blocks,
                          void *d_out, int check_crc, int loop, uint64_t out_bytes, const WaveBufs *wb = nullptr) {
    if (!n_blocks) return PSSBAM_OK;
    if (!d_comp || !d_blocks || !d_out) return fail(PSSBAM_EINVAL, "null buffer");
    if ((uintptr_t)d_comp & 3u) return fail(PSSBAM_EINVAL, "d_comp must be 4-byte aligned");
    int dev = 0, n_cu = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    static bool attr_set[64] = {false};
    if (!attr_set[dev & 63]) {
        HIP_TRY(hipFuncSetAttribute((const void *)pssbam::bgzf_inflate_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pssbam::INF_LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute((const void *)pssbam::bgzf_inflate_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pssbam::INF_LDS_BYTES_DEFER));
        HIP_TRY(hipFuncSetAttribute((const void *)pssbam::bgzf_inflate_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pssbam::INF_LDS_BYTES_DEFER));
        HIP_TRY(hipFuncSetAttribute((const void *)pssbam::bgzf_inflate_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pssbam::INF_LDS_BYTES_DEFER));
        HIP_TRY(hipFuncSetAttribute((const void *)pssbam::bgzf_inflate_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pssbam::INF_LDS_BYTES_DEFER));
        attr_set[dev & 63] = true;
    }
    if (loop == -2) loop = 0;   // (not negotiable)
    else if (const char *lm = getenv("PSSBAM_INFLATE_LOOP")) loop = atoi(lm) != 0;
    if (comp_bytes < 64u) loop = 0;
    const uint32_t groups = (n_blocks + pssbam::INF_WAVE - 1) / pssbam::INF_WAVE;
    const char *gm = getenv("PSSBAM_INFLATE_WAVES_PER_CU");
    const uint32_t per_cu = gm && atoi(gm) > 0 ? (uint32_t)atoi(gm) : (uint32_t)pssbam::INF_WAVES_PER_CU;
    const uint32_t grid = std::min<uint32_t>(groups, (uint32_t)n_cu * per_cu);
    // match-dominated batches (a BAM with constant or heavily binned QUAL inflates 8x and more) do best with short
    // literal runs, literal-dominated ones with longer ones
    // (round 3, after the decode got cheaper: 40-level QUAL 118.8 / 134.1 / 140.4 GB/s for 4 / 6 / 8, four-bin 194 either way)
    uint32_t lit_run = out_bytes && out_bytes >= 8ull * comp_bytes ? 4u : 8u;
    if (const char *lr = getenv("PSSBAM_INFLATE_RUN")) lit_run = (uint32_t)std::max(1, atoi(lr));
    const char *pv = getenv("PSSBAM_INFLATE_PIECES");
    const bool pieces = pv ? atoi(pv) != 0 : true;   // bounded work per lane and step (csrc/inflate_kernels.h INF_PIECE)
    uint32_t only_status = 0xFFFFFFFFu;
    if (wb && loop > 0 && pieces && wave_inflate_wanted() && !getenv("PSSBAM_INFLATE_STAMPS")) {
        // one wave per block: Huffman decoding into sequences + literals, then LZ77 in a 64 KiB LDS window per block; the
        // lane-per-block kernel below only takes the blocks that path hands back (INF_RETRY)
        const char *wg = getenv("PSSBAM_WAVE_WGS_PER_CU");
        const uint32_t tgrid = std::min<uint32_t>((n_blocks + pssbam::WV_WAVES - 1) / pssbam::WV_WAVES, (uint32_t)n_cu * (wg && atoi(wg) > 0 ? (uint32_t)atoi(wg) : 12u));   // (13 KiB of LDS per one-wave workgroup: twelve per CU)
        unsigned long long *d_wdbg = nullptr;
        if (getenv("PSSBAM_WAVE_STAMPS")) {   // diagnostics: where the token kernel's cycles go
            static unsigned long long *d_dbg2 = nullptr;
            if (!d_dbg2) HIP_TRY(hipMalloc(&d_dbg2, 20 * sizeof(unsigned long long)));
            HIP_TRY(hipMemsetAsync(d_dbg2, 0, 20 * sizeof(unsigned long long), st));
            d_wdbg = d_dbg2;
        }
        hipLaunchKernelGGL(pssbam::bgzf_tokens_kernel, dim3(tgrid), dim3(64 * pssbam::WV_WAVES), 0, st, (const uint8_t *)d_comp, comp_bytes, (pssbam::BgzfBlock *)d_blocks,
                           n_blocks, (uint8_t *)d_out, wb->seq_arena, wb->seq_off, wb->seq_count, d_wdbg);
        const char *rg = getenv("PSSBAM_RESOLVE_WGS_PER_CU");
        const uint32_t rgrid = std::min<uint32_t>((n_blocks + pssbam::RS_WAVES - 1) / pssbam::RS_WAVES, (uint32_t)n_cu * (rg && atoi(rg) > 0 ? (uint32_t)atoi(rg) : 4u));
        hipLaunchKernelGGL(pssbam::bgzf_resolve_kernel, dim3(rgrid), dim3(64 * pssbam::RS_WAVES), 0, st, (uint8_t *)d_out, (pssbam::BgzfBlock *)d_blocks, n_blocks,
                           (const uint32_t *)wb->seq_arena, wb->seq_off, (const uint32_t *)wb->seq_count, d_wdbg);
        HIP_TRY(hipGetLastError());
        if (d_wdbg) {
            unsigned long long h[20];
            HIP_TRY(hipMemcpyAsync(h, d_wdbg, sizeof h, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            const double nb = (double)std::max<unsigned long long>(h[9], 1), rb = (double)std::max<unsigned long long>(h[16], 1);
            fprintf(stderr, "[pssbam] token kernel stamps, shader cycles per BGZF block (%.0f blocks, %.2f chunks each): set-up+stored %.0f, code-length walk %.0f, table builds %.0f, "
                            "walk A1 %.0f, A2 %.0f, path %.0f, B+scans %.0f, C %.0f\n", nb, h[8] / nb, h[0] / nb, h[1] / nb, h[2] / nb, h[3] / nb, h[4] / nb, h[5] / nb, h[6] / nb, h[7] / nb);
            fprintf(stderr, "[pssbam] resolve kernel stamps, shader cycles per BGZF block (%.0f blocks, %.1f batches and %.1f rounds each): window in %.0f, scans %.0f, rounds %.0f, window out %.0f\n",
                    rb, h[14] / rb, h[15] / rb, h[10] / rb, h[11] / rb, h[12] / rb, h[13] / rb);
        }
        only_status = pssbam::INF_RETRY;
    }
    if (loop > 0 && getenv("PSSBAM_INFLATE_STAMPS")) {   // diagnostic build: where the cycles of a step go (tools/inflate_stamps.sh)
        static unsigned long long *d_dbg = nullptr;
        if (!d_dbg) HIP_TRY(hipMalloc(&d_dbg, 8 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(d_dbg, 0, 8 * sizeof(unsigned long long), st));
        if (pieces)
            hipLaunchKernelGGL((pssbam::bgzf_inflate_kernel<true, true, true>), dim3(grid), dim3(pssbam::INF_WAVE), pssbam::INF_LDS_BYTES_DEFER, st, (const uint8_t *)d_comp,
                               comp_bytes, (pssbam::BgzfBlock *)d_blocks, n_blocks, (uint8_t *)d_out, lit_run, d_dbg);
        else
            hipLaunchKernelGGL((pssbam::bgzf_inflate_kernel<true, false, true>), dim3(grid), dim3(pssbam::INF_WAVE), pssbam::INF_LDS_BYTES_DEFER, st, (const uint8_t *)d_comp,
                               comp_bytes, (pssbam::BgzfBlock *)d_blocks, n_blocks, (uint8_t *)d_out, lit_run, d_dbg);
        unsigned long long h[8];
        HIP_TRY(hipMemcpyAsync(h, d_dbg, sizeof h, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const double L = (double)std::max<unsigned long long>(h[3], 1);
        fprintf(stderr, "[pssbam] inflate stamps (pieces %d; lane sums, shader cycles): steps %llu, per step: decode %.0f wait %.0f stores+copies %.0f cycles; lanes in a step %.1f of 64; "
                        "literals per step %.2f; steps ending in an in-place copy %.3f; in-loop share of the lanes' kernel cycles %.3f\n", (int)pieces,
                h[3], h[0] / L, h[1] / L, h[2] / L, h[4] / L, h[5] / L, h[6] / L, (double)(h[0] + h[1] + h[2]) / (double)std::max<unsigned long long>(h[7], 1));
    } else if (only_status != 0xFFFFFFFFu && getenv("PSSBAM_INFLATE_WAVE") && atoi(getenv("PSSBAM_INFLATE_WAVE")) == 2) {
        // (diagnostics: no second chance -- a block the wave path handed back stays in state INF_RETRY and is reported)
    } else if (loop > 0 && pieces)
        hipLaunchKernelGGL((pssbam::bgzf_inflate_kernel<true, true>), dim3(grid), dim3(pssbam::INF_WAVE), pssbam::INF_LDS_BYTES_DEFER, st, (const uint8_t *)d_comp,
                           comp_bytes, (pssbam::BgzfBlock *)d_blocks, n_blocks, (uint8_t *)d_out, lit_run, (unsigned long long *)nullptr, only_status);
    else if (loop > 0)
        hipLaunchKernelGGL(pssbam::bgzf_inflate_kernel<true>, dim3(grid), dim3(pssbam::INF_WAVE), pssbam::INF_LDS_BYTES_DEFER, st, (const uint8_t *)d_comp,
                           comp_bytes, (pssbam::BgzfBlock *)d_blocks, n_blocks, (uint8_t *)d_out, lit_run);
    else
        hipLaunchKernelGGL(pssbam::bgzf_inflate_kernel<false>, dim3(grid), dim3(pssbam::INF_WAVE), pssbam::INF_LDS_BYTES, st, (const uint8_t *)d_comp,
                           comp_bytes, (pssbam::BgzfBlock *)d_blocks, n_blocks, (uint8_t *)d_out, lit_run);
    HIP_TRY(hipGetLastError());
    if (check_crc) return launch_crc(st, d_blocks, n_blocks, d_out);
    return PSSBAM_OK;
}

extern "C" int pssbam_bgzf_inflate_device(void *hip_stream, const void *d_comp, uint64_t comp_bytes, pssbam_bgzf_block *d_blocks,
                                          uint32_t n_blocks, void *d_out, int check_crc) {
    return launch_inflate((hipStream_t)hip_stream, d_comp, comp_bytes, d_blocks, n_blocks, d_out, check_crc, -2, 0);   // a caller's buffers: no over-reads
}


// Convenience for tests and tools: host BGZF bytes in -> inflated bytes out (host), everything in
// between on the device.  *kernel_ms = device time of the inflate (+ CRC) kernels alone.
extern "C" int pssbam_bgzf_inflate_host(int device, const void *bgzf, uint64_t nbytes, void *out, uint64_t out_cap, uint64_t *out_len,
                                        uint32_t *n_blocks_out, uint32_t *first_bad_block, uint32_t *first_bad_status, double *kernel_ms,
                                        int check_crc, int repeats) {
    uint64_t consumed = 0, total = 0;
    const int64_t n = pssbam_bgzf_scan(bgzf, nbytes, nullptr, 0, &consumed, &total);
    if (n < 0) return (int)n;
    if (consumed != nbytes) return fail(PSSBAM_EFORMAT, "input ends inside a BGZF block");
    if (out_len) *out_len = total;
    if (n_blocks_out) *n_blocks_out = (uint32_t)n;
    if (first_bad_block) *first_bad_block = 0xFFFFFFFFu;
    if (first_bad_status) *first_bad_status = 0;
    if (n == 0) return PSSBAM_OK;
    if (out && total > out_cap) return fail(PSSBAM_EINVAL, "output buffer too small (%llu needed)", (unsigned long long)total);
    if (n > 0xFFFFFFF0ll) return fail(PSSBAM_EINVAL, "too many blocks");
    std::vector<pssbam_bgzf_block> blocks((size_t)n);
    (void)pssbam_bgzf_scan(bgzf, nbytes, blocks.data(), (uint64_t)n, nullptr, nullptr);
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    uint8_t *d_comp = nullptr, *d_out = nullptr;
    pssbam_bgzf_block *d_blocks = nullptr;
    uint32_t *d_seq = nullptr, *d_seq_count = nullptr;
    uint64_t *d_seq_off = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = PSSBAM_OK;
    auto cleanup = [&]() {
        if (d_comp) (void)hipFree(d_comp);
        if (d_out) (void)hipFree(d_out);
        if (d_blocks) (void)hipFree(d_blocks);
        if (d_seq) (void)hipFree(d_seq);
        if (d_seq_off) (void)hipFree(d_seq_off);
        if (d_seq_count) (void)hipFree(d_seq_count);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
#define TRY_C(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return fail(PSSBAM_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e)); } } while (0)
    TRY_C(hipMalloc(&d_comp, nbytes + 64));
    std::vector<uint64_t> seq_off((size_t)n + 1, 0);
    for (int64_t i = 0; i < n; i++) seq_off[(size_t)i + 1] = seq_off[(size_t)i] + pssbam::seq_cap_of(blocks[(size_t)i].isize);
    TRY_C(hipMalloc(&d_seq, (seq_off[(size_t)n] + 64) * sizeof(uint32_t)));
    TRY_C(hipMalloc(&d_seq_off, ((size_t)n + 1) * sizeof(uint64_t)));
    TRY_C(hipMalloc(&d_seq_count, (size_t)n * sizeof(uint32_t)));
    TRY_C(hipMemcpy(d_seq_off, seq_off.data(), ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    const WaveBufs wbufs{d_seq, d_seq_off, d_seq_count};
    TRY_C(hipMalloc(&d_out, total + 128));   // (slack for the one-wait-per-step loop's requests: up to 64 bytes past a block's end)
    TRY_C(hipMalloc(&d_blocks, (size_t)n * sizeof(pssbam_bgzf_block)));
    TRY_C(hipMemset(d_comp + nbytes, 0, 64));
    TRY_C(hipMemcpy(d_comp, bgzf, nbytes, hipMemcpyHostToDevice));
    TRY_C(hipMemcpy(d_blocks, blocks.data(), (size_t)n * sizeof(pssbam_bgzf_block), hipMemcpyHostToDevice));
    TRY_C(hipEventCreate(&e0));
    TRY_C(hipEventCreate(&e1));
    if (repeats < 1) repeats = 1;
    float best = 1e30f;
    for (int r = 0; r < repeats && rc == PSSBAM_OK; r++) {
        TRY_C(hipEventRecord(e0, nullptr));
        rc = launch_inflate(nullptr, d_comp, nbytes, d_blocks, (uint32_t)n, d_out, check_crc, INFLATE_LOOP_DEFAULT, total, &wbufs);
        if (rc) break;
        TRY_C(hipEventRecord(e1, nullptr));
        TRY_C(hipEventSynchronize(e1));
        float ms = 0.f;
        TRY_C(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    if (rc) { cleanup(); return rc; }
    if (kernel_ms) *kernel_ms = best;
    TRY_C(hipMemcpy(blocks.data(), d_blocks, (size_t)n * sizeof(pssbam_bgzf_block), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; i++)
        if (blocks[(size_t)i].status) {
            if (first_bad_block) *first_bad_block = (uint32_t)i;
            if (first_bad_status) *first_bad_status = blocks[(size_t)i].status;
            break;
        }
    if (out) TRY_C(hipMemcpy(out, d_out, total, hipMemcpyDeviceToHost));
#undef TRY_C
    cleanup();
    return PSSBAM_OK;
}

// --------------------------------------------------------------------------------------
// the feed: compressed chunks -> device inflate -> device record index -> tally
// --------------------------------------------------------------------------------------
// The inflate kernel runs one lane per BGZF block, 65 536 lanes resident: it wants whole "rounds" of that many
// blocks per launch (a 1 GiB batch holds 16 000 blocks).  submit_bgzf therefore only COPIES its chunk and
// appends its blocks to the super-batch being assembled; when that holds one round (4.3 GB of output) it is
// flushed: one inflate launch + one CRC launch over all blocks, then per < 4 GiB sub-batch (the tally kernels
// index records with u32 offsets) the record index and the tally.  Super-batches live in a ring of slots, so
// chunks keep arriving over PCIe while earlier ones are inflated -- and, after pssbam_engine_feed_open, while
// the GENOME is still on its way: inflate, CRC and record index need no reference base, so they run at once and
// only the tally launches of a super-batch are put off until set_genome + set_references have been called (the
// ring grows meanwhile, within what the device has free).

template <class T>
static int grow(T **ptr, size_t *cap, size_t need, size_t elem = sizeof(T)) {
    if (*cap >= need) return PSSBAM_OK;
    if (*ptr) HIP_TRY(hipFree(*ptr));
    *ptr = nullptr;
    *cap = need + need / 8 + 4096;
    HIP_TRY(hipMalloc((void **)ptr, *cap * elem));
    return PSSBAM_OK;
}

static constexpr uint64_t FEED_SUB_MAX = (3584ull << 20);   // records per tally launch: below 4 GiB

// Device buffers of the feed can be reserved ahead of the first submit (pssbam_feed_reserve, from a
// helper thread while the caller still loads its FASTA): allocating ~13 GB takes from a millisecond
// to a second depending on what the driver has to reclaim.  What no engine took goes back with
// pssbam_feed_release.
namespace {
struct FeedReserve {
    uint8_t *comp[FEED_SLOTS_READY] = {nullptr}, *out[FEED_SLOTS_READY] = {nullptr};
    size_t comp_cap = 0, out_cap = 0;
    int pending = 0;   // pairs a pssbam_feed_reserve call in progress has yet to publish
};
FeedReserve g_feed_reserve[64];
std::mutex g_feed_reserve_mu;
std::condition_variable g_feed_reserve_cv;
}  // namespace

// Allocates the buffers of the feed's first slots for `device` and publishes them one by one (an engine that needs a
// slot meanwhile waits for the next one instead of allocating beside this thread: device allocations of this size take
// from a millisecond to hundreds of ms each, depending on what the driver has to clear -- profiles/r03_exit_teardown_probe.txt).
extern "C" int pssbam_feed_reserve(int device) {
    if (device < 0 || device >= 64) return fail(PSSBAM_EINVAL, "device %d out of range", device);
    {
        std::lock_guard<std::mutex> lk(g_feed_reserve_mu);
        FeedReserve &r = g_feed_reserve[device];
        if (r.pending) return PSSBAM_OK;   // somebody is at it
        for (int k = 0; k < FEED_SLOTS_READY; k++)
            if (r.comp[k]) return PSSBAM_OK;   // ... or has been
        r.comp_cap = (size_t)FEED_COMP_CAP;
        r.out_cap = (size_t)(FEED_GAP + FEED_OUT_TARGET + FEED_OUT_SLACK);
        r.pending = FEED_SLOTS_READY;
    }
    int rc = PSSBAM_OK;
    hipError_t err = hipSetDevice(device);
    for (int k = 0; k < FEED_SLOTS_READY; k++) {
        uint8_t *c = nullptr, *o = nullptr;
        if (err == hipSuccess) err = hipMalloc(&c, (size_t)FEED_COMP_CAP + 64);
        if (err == hipSuccess) err = hipMalloc(&o, (size_t)(FEED_GAP + FEED_OUT_TARGET + FEED_OUT_SLACK));
        std::lock_guard<std::mutex> lk(g_feed_reserve_mu);
        FeedReserve &r = g_feed_reserve[device];
        if (err == hipSuccess) { r.comp[k] = c; r.out[k] = o; }
        else if (c) (void)hipFree(c);
        r.pending--;
        g_feed_reserve_cv.notify_all();
    }
    if (err != hipSuccess) rc = fail(PSSBAM_EHIP, "reserving feed buffers failed: %s", hipGetErrorString(err));
    return rc;
}

extern "C" int pssbam_feed_release(int device) {
    if (device < 0 || device >= 64) return fail(PSSBAM_EINVAL, "device %d out of range", device);
    FeedReserve r;
    {
        std::unique_lock<std::mutex> lk(g_feed_reserve_mu);
        g_feed_reserve_cv.wait(lk, [&] { return g_feed_reserve[device].pending == 0; });
        r = g_feed_reserve[device];
        g_feed_reserve[device] = FeedReserve();
    }
    bool any = false;
    for (int k = 0; k < FEED_SLOTS_READY; k++) any = any || r.comp[k];
    if (!any) return PSSBAM_OK;
    HIP_TRY(hipSetDevice(device));
    for (int k = 0; k < FEED_SLOTS_READY; k++) {
        if (r.comp[k]) (void)hipFree(r.comp[k]);
        if (r.out[k]) (void)hipFree(r.out[k]);
    }
    return PSSBAM_OK;
}

// takes one (comp, out) pair from the device's reserve, if there is (or is about to be) one of the wanted size
static bool feed_take_reserved(int device, size_t comp_cap, size_t out_need, uint8_t **comp, uint8_t **out, size_t *out_cap) {
    if (device < 0 || device >= 64) return false;
    std::unique_lock<std::mutex> lk(g_feed_reserve_mu);
    FeedReserve &r = g_feed_reserve[device];
    for (;;) {
        if (r.comp_cap != comp_cap || r.out_cap < out_need) return false;
        for (int k = 0; k < FEED_SLOTS_READY; k++)
            if (r.comp[k]) {
                *comp = r.comp[k];
                *out = r.out[k];
                *out_cap = r.out_cap;
                r.comp[k] = r.out[k] = nullptr;
                return true;
            }
        if (!r.pending) return false;
        g_feed_reserve_cv.wait(lk);
    }
}

static bool feed_engine_ready(const pssbam_engine *e) { return e->have_refs; }   // (set_references needs the genome: implied)

// what one slot of the ring costs in device memory (buffers + offset index), for the budget
static uint64_t feed_slot_bytes(const pssbam_engine *e) {
    const uint64_t out = FEED_GAP + e->feed_out_target + FEED_OUT_SLACK;
    return out + e->feed_comp_cap + out / 8 + (64ull << 20);
}

// buffers of a super-batch that do not depend on its contents
static int feed_prepare(pssbam_engine *e, FeedAcc &s) {
    const size_t out_need = (size_t)(FEED_GAP + e->feed_out_target + FEED_OUT_SLACK);
    if (!s.d_comp) {
        s.comp_cap = (size_t)e->feed_comp_cap;
        if (!feed_take_reserved(e->device, s.comp_cap, out_need, &s.d_comp, &s.d_out, &s.out_cap)) HIP_TRY(hipMalloc(&s.d_comp, s.comp_cap + 64));
    }
    if (!s.d_out) {
        int rc = grow(&s.d_out, &s.out_cap, out_need);
        if (rc) return rc;
    }
    if (!s.d_chain) {
        HIP_TRY(hipMalloc(&s.d_chain, 4 * sizeof(uint64_t)));
        HIP_TRY(hipMemsetAsync(s.d_chain, 0, 4 * sizeof(uint64_t), e->stream));
    }
    if (!s.consumed) HIP_TRY(hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming));
    return PSSBAM_OK;
}

// A slot to assemble the next super-batch in.  Order of preference: one that is free, one whose kernels have
// completed, a new one (three while the engine is ready, as many as the memory budget allows while tallies are put
// off), the oldest one still running (wait for it).  PSSBAM_EBUSY: every slot is full of inflated records that wait
// for the genome -- unless `force`, which then goes past the budget.
static int feed_acquire(pssbam_engine *e, int *out_slot, bool force) {
    int oldest = -1;
    for (size_t i = 0; i < e->feed.size(); i++) {
        FeedAcc &s = *e->feed[i];
        if ((int)i == e->cur_feed || (int)i == e->spare_feed) continue;
        if (!s.busy) { *out_slot = (int)i; return PSSBAM_OK; }
        if (!s.held && (oldest < 0 || s.flush_seq < e->feed[(size_t)oldest]->flush_seq)) oldest = (int)i;
    }
    if (oldest >= 0 && hipEventQuery(e->feed[(size_t)oldest]->consumed) == hipSuccess) {
        e->feed[(size_t)oldest]->busy = false;
        *out_slot = oldest;
        return PSSBAM_OK;
    }
    (void)hipGetLastError();   // (hipErrorNotReady is not an error)
    size_t limit = FEED_SLOTS_READY;
    if (const char *v = getenv("PSSBAM_FEED_SLOTS")) limit = std::max<size_t>(2, (size_t)atoi(v));   // (experiments: the ring's depth once the genome is set)
    if (!feed_engine_ready(e)) {
        if (!e->feed_mem_budget) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 64ull << 30;
            const uint64_t used = e->feed.size() * feed_slot_bytes(e);
            e->feed_mem_budget = used + (free_b > (24ull << 30) ? free_b - (24ull << 30) : 0);   // leaves room for genome, 4-bit image, k-mer bins
        }
        limit = std::max<size_t>(FEED_SLOTS_READY, (size_t)(e->feed_mem_budget / feed_slot_bytes(e)));
        if (const char *v = getenv("PSSBAM_FEED_MAX_SLOTS")) limit = std::max<size_t>(1, (size_t)atoi(v));
    }
    limit = std::min<size_t>(limit, FEED_SLOTS_MAX);
    if (e->feed.size() < limit || (force && oldest < 0 && e->feed.size() < (size_t)FEED_SLOTS_MAX)) {
        const double t0 = feed_now();
        FeedAcc *s = new FeedAcc();
        e->feed.push_back(s);
        const int rc = feed_prepare(e, *s);
        e->feed_t_alloc += feed_now() - t0;
        if (rc) return rc;
        e->feed_slots_allocated++;
        *out_slot = (int)e->feed.size() - 1;
        return PSSBAM_OK;
    }
    if (oldest >= 0) {
        const double t0 = feed_now();
        HIP_TRY(hipEventSynchronize(e->feed[(size_t)oldest]->consumed));
        e->feed_t_wait_busy += feed_now() - t0;
        e->feed[(size_t)oldest]->busy = false;
        *out_slot = oldest;
        return PSSBAM_OK;
    }
    return fail(PSSBAM_EBUSY, "all %zu feed slots hold inflated records that wait for set_genome / set_references", e->feed.size());
}

// the tally launches of one flushed super-batch; `consumed` is recorded behind the last of them
static int feed_launch_tally(pssbam_engine *e, const DeferredTally &d) {
    FeedAcc &s = *e->feed[(size_t)d.slot];
    int rc = launch_tally(e, s.d_out + d.sub_base, d.sub_len, s.d_offs + d.offs_at, d.n_bound, nullptr, 0, s.d_nrecs + d.k, d.sample_off);
    if (rc) return rc;
    if (d.last_of_slot) {
        HIP_TRY(hipEventRecord(s.consumed, e->stream));
        s.held = false;
    }
    return PSSBAM_OK;
}

// set_genome + set_references have both been called: the super-batches inflated ahead of them are tallied
static int feed_resume(pssbam_engine *e) {
    if (e->deferred.empty() || !feed_engine_ready(e)) return PSSBAM_OK;
    HIP_TRY(hipSetDevice(e->device));
    std::vector<DeferredTally> todo;
    todo.swap(e->deferred);
    for (const DeferredTally &d : todo) {
        const int rc = feed_launch_tally(e, d);
        if (rc) return rc;
        e->feed_deferred_launches++;
    }
    return PSSBAM_OK;
}

static int feed_flush(pssbam_engine *e) {
    if (e->cur_feed < 0) return PSSBAM_OK;
    FeedAcc &s = *e->feed[(size_t)e->cur_feed];
    if (s.blocks.empty()) return PSSBAM_OK;
    int rc;
    const double t_flush0 = feed_now();
    if (e->feed_t_first_flush < 0) e->feed_t_first_flush = t_flush0 - e->feed_t0;
    struct FlushTimer { pssbam_engine *e; double t0; ~FlushTimer() { e->feed_t_flush += feed_now() - t0; } } flush_timer{e, t_flush0};
    const size_t nb = s.blocks.size();
    if (nb > 0xFFFFFFF0ull) return fail(PSSBAM_EINVAL, "too many BGZF blocks in one super-batch");
    if (s.blocks_cap < nb) {
        const size_t want = std::max<size_t>(nb, e->feed_block_target < 0xFFFFFFFFull ? (size_t)e->feed_block_target : 0);
        size_t c[8] = {s.blocks_cap, s.blocks_cap, s.blocks_cap, s.blocks_cap, s.blocks_cap, s.blocks_cap, s.blocks_cap, s.blocks_cap};
        if ((rc = grow((uint8_t **)&s.d_blocks, &c[0], want, sizeof(pssbam::BgzfBlock)))) return rc;
        if ((rc = grow(&s.d_a, &c[1], want))) return rc;
        if ((rc = grow(&s.d_e, &c[2], want))) return rc;
        if ((rc = grow(&s.d_last, &c[3], want))) return rc;
        if ((rc = grow(&s.d_nexta, &c[4], want + 1))) return rc;
        if ((rc = grow(&s.d_n, &c[5], want))) return rc;
        if ((rc = grow(&s.d_counts, &c[6], want))) return rc;
        if ((rc = grow(&s.d_base, &c[7], want))) return rc;
        s.blocks_cap = *std::min_element(c, c + 8);
    }
    const uint64_t data_end = s.out_used;   // the blocks sit contiguously in [FEED_GAP, data_end)
    // (offsets: a block of isize bytes starts at most isize / 36 + 1 records, whatever its bytes are)
    if ((rc = grow(&s.d_offs, &s.offs_cap, (size_t)(std::max<uint64_t>(data_end, FEED_GAP + e->feed_out_target) / 36ull + std::max<size_t>(nb, s.blocks_cap) + 2ull * s.sub_first.size() + 64ull)))) return rc;
    if ((rc = grow(&s.d_nrecs, &s.nrecs_cap, std::max<size_t>(s.sub_first.size(), 4)))) return rc;
    if (!e->d_carry) HIP_TRY(hipMalloc(&e->d_carry, FEED_GAP));
    // every chunk of this super-batch has been issued on the copy streams: the engine's stream waits for them
    if (!s.copies_done) {
        HIP_TRY(hipEventCreateWithFlags(&s.copies_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.copies_done2, hipEventDisableTiming));
    }
    late_streams(e, false);
    // (chunks that went out before the copy streams existed were copied on the engine's stream: in order with what follows)
    HIP_TRY(hipEventRecord(s.copies_done, e->copy_stream ? e->copy_stream : e->stream));
    HIP_TRY(hipEventRecord(s.copies_done2, e->copy_stream2 ? e->copy_stream2 : e->copy_stream ? e->copy_stream : e->stream));
    // The inflate launch can go to one of two streams of its own (PSSBAM_FEED_INFLATE_STREAMS=2: launches take turns, the head
    // of one fills the CUs the tail of the previous one leaves; CRC / index / tally stay on the engine's stream behind an
    // event).  ✗ Measured on the 200 M-read command: feed phase 0.340-0.362 s with, 0.337-0.363 s without -- the blocks of a
    // launch finish close together and the launches were back to back already -- so one queue remains the default.
    hipStream_t is = e->stream;
    if (const char *v = getenv("PSSBAM_FEED_INFLATE_STREAMS")) {
        if (atoi(v) >= 2) {
            for (hipStream_t &q : e->inflate_stream)
                if (!q) HIP_TRY(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));   // (made on first use: a stream costs milliseconds at start-up)
            is = e->inflate_stream[e->flush_seq & 1u];
        }
    }
    HIP_TRY(hipStreamWaitEvent(is, s.copies_done, 0));
    HIP_TRY(hipStreamWaitEvent(is, s.copies_done2, 0));
    // the table goes up from page-locked memory: a pageable source makes hipMemcpyAsync wait for everything queued on
    // the stream (the previous super-batch's kernels), and this thread has the next super-batch's copies to issue
    if (s.h_blocks_cap < nb) {
        if (s.h_blocks) (void)hipHostFree(s.h_blocks);
        s.h_blocks = nullptr;
        s.h_blocks_cap = 0;
        const size_t cap = std::max<size_t>(nb + nb / 4, 1u << 16);
        if (hipHostMalloc((void **)&s.h_blocks, cap * sizeof(pssbam_bgzf_block), hipHostMallocDefault) == hipSuccess) s.h_blocks_cap = cap;
        else s.h_blocks = nullptr;   // (pageable then: slower, not wrong)
    }
    const pssbam_bgzf_block *table = s.blocks.data();
    if (s.h_blocks) {
        memcpy(s.h_blocks, s.blocks.data(), nb * sizeof(pssbam_bgzf_block));
        table = s.h_blocks;
    }
    HIP_TRY(hipMemcpyAsync(s.d_blocks, table, nb * sizeof(pssbam_bgzf_block), hipMemcpyHostToDevice, is));
    hipEvent_t ev0 = take_event(e), ev1 = take_event(e);
    if (!ev0 || !ev1) return fail(PSSBAM_EHIP, "hipEventCreate failed");
    if (!e->feed_base_ev) {
        HIP_TRY(hipEventCreate(&e->feed_base_ev));
        HIP_TRY(hipEventRecord(e->feed_base_ev, is));
    }
    HIP_TRY(hipEventRecord(ev0, is));   // (behind the table's copy: the interval is kernels only, not the copy's wait for a DMA engine)
    rc = launch_inflate(is, s.d_comp, s.comp_used, (pssbam_bgzf_block *)s.d_blocks, (uint32_t)nb, s.d_out, 0, INFLATE_LOOP_DEFAULT, data_end - FEED_GAP);
    if (rc) return rc;
    if (!s.inflate_done) HIP_TRY(hipEventCreateWithFlags(&s.inflate_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(s.inflate_done, is));
    HIP_TRY(hipStreamWaitEvent(e->stream, s.inflate_done, 0));
    if (e->feed_fresh) {   // the stream's first super-batch: its chain starts behind the BAM header
        const uint64_t first = FEED_GAP + e->feed_skip;
        HIP_TRY(hipMemcpyAsync(s.d_chain, &first, sizeof first, hipMemcpyHostToDevice, e->stream));   // (pageable source: copied before the call returns)
        e->feed_fresh = false;
    } else {               // the partial record the previous super-batch ended with goes in front of this one's data
        hipLaunchKernelGGL(pssbam::bgzf_chain_carry_in, dim3(1), dim3(256), 0, e->stream, (const uint8_t *)e->d_carry, (const uint64_t *)e->d_feed_tail,
                           s.d_out, (uint64_t)FEED_GAP, s.d_chain);
    }
    if (!getenv("PSSBAM_NO_CRC")) {
        rc = launch_crc(e->stream, (pssbam_bgzf_block *)s.d_blocks, (uint32_t)nb, s.d_out, is != e->stream && !getenv("PSSBAM_FEED_BIG_CRC"));
        if (rc) return rc;
    }
    // the record chain of the whole super-batch: per-block pieces, linked and checked
    const int32_t n_ref = e->have_refs ? e->n_ref : e->feed_n_ref;
    {
        const uint32_t n = (uint32_t)nb, grid = std::min<uint32_t>((n + 255u) / 256u, (uint32_t)e->n_cu * 8u);
        const pssbam::BgzfBlock *blk = (const pssbam::BgzfBlock *)s.d_blocks;
        hipLaunchKernelGGL(pssbam::bgzf_chain_spec, dim3(grid), dim3(256), 0, e->stream, (const uint8_t *)s.d_out, blk, n, data_end,
                           (const uint64_t *)s.d_chain, n_ref, s.d_a, s.d_n, s.d_e, s.d_last, e->d_feed_flags, s.d_chain + 2);
        hipLaunchKernelGGL(pssbam::bgzf_chain_suffix, dim3(1), dim3(1024), 0, e->stream, (const uint64_t *)s.d_a, n, s.d_nexta, (const uint64_t *)nullptr);
        hipLaunchKernelGGL(pssbam::bgzf_chain_verify<false>, dim3(grid), dim3(256), 0, e->stream, (const uint8_t *)s.d_out, (const uint32_t *)s.d_n, (const uint64_t *)s.d_e,
                           (const uint64_t *)s.d_last, (const uint64_t *)s.d_nexta, n, data_end, (const uint64_t *)s.d_chain, s.d_counts,
                           s.d_chain + 1, e->d_feed_flags, s.d_chain + 2);
        // links that do not hold are repaired from the left (a serial walk over the blocks in doubt only); what was
        // repaired is linked and checked again.  Without a broken link the three launches return at once.
        const char *rv = getenv("PSSBAM_FEED_REPAIR");
        const int repair = rv ? atoi(rv) : 1;
        hipLaunchKernelGGL(pssbam::bgzf_chain_repair, dim3(1), dim3(64), 0, e->stream, (const uint8_t *)s.d_out, blk, n, data_end, (const uint64_t *)s.d_chain,
                           n_ref, s.d_a, s.d_n, s.d_e, s.d_last, (const uint64_t *)s.d_nexta, e->d_feed_flags, s.d_chain + 2, repair);
        hipLaunchKernelGGL(pssbam::bgzf_chain_suffix, dim3(1), dim3(1024), 0, e->stream, (const uint64_t *)s.d_a, n, s.d_nexta, (const uint64_t *)(s.d_chain + 3));
        hipLaunchKernelGGL(pssbam::bgzf_chain_verify<true>, dim3(grid), dim3(256), 0, e->stream, (const uint8_t *)s.d_out, (const uint32_t *)s.d_n, (const uint64_t *)s.d_e,
                           (const uint64_t *)s.d_last, (const uint64_t *)s.d_nexta, n, data_end, (const uint64_t *)s.d_chain, s.d_counts,
                           s.d_chain + 1, e->d_feed_flags, s.d_chain + 2);
    }
    // offsets per tally sub-batch: the records STARTING in its blocks, relative to a 16-byte aligned base
    uint64_t offs_at = 0;
    std::vector<uint64_t> sub_offs(s.sub_first.size()), sub_base(s.sub_first.size()), sub_len(s.sub_first.size());
    for (size_t k = 0; k < s.sub_first.size(); k++) {
        const uint32_t b0 = s.sub_first[k], b1 = k + 1 < s.sub_first.size() ? s.sub_first[k + 1] : (uint32_t)nb;
        // (sub-batch 0 owns the carried record in the gap; a later one may be entered by the last record of its predecessor)
        const uint64_t base = k == 0 ? 0ull : (s.blocks[b0].out_off & ~15ull);
        const uint64_t end = std::min<uint64_t>(data_end, s.blocks[b1 - 1].out_off + s.blocks[b1 - 1].isize + FEED_GAP);
        sub_offs[k] = offs_at;
        sub_base[k] = base;
        sub_len[k] = end - base;
        const uint32_t n = b1 - b0;
        const uint32_t igrid = std::min<uint32_t>((n + 255u) / 256u, (uint32_t)e->n_cu * 8u);
        hipLaunchKernelGGL(pssbam::bgzf_index_scan, dim3(1), dim3(1024), 0, e->stream, (const uint32_t *)(s.d_counts + b0), n, s.d_base + b0,
                           s.d_nrecs + k);
        hipLaunchKernelGGL(pssbam::bgzf_chain_write, dim3(igrid), dim3(256), 0, e->stream, (const uint8_t *)s.d_out, (const uint64_t *)(s.d_a + b0),
                           (const uint32_t *)(s.d_counts + b0), (const uint32_t *)(s.d_base + b0), n, base, s.d_offs + offs_at,
                           (const uint32_t *)(s.d_nrecs + k));
        offs_at += sub_len[k] / 36ull + n + 2ull;
    }
    // the partial record at the end sets out for the next super-batch (whichever slot that will be)
    hipLaunchKernelGGL(pssbam::bgzf_chain_carry_out, dim3(1), dim3(256), 0, e->stream, (const uint8_t *)s.d_out, (const uint64_t *)(s.d_chain + 1), data_end,
                       e->d_carry, (uint64_t)FEED_GAP, e->d_feed_tail, e->d_feed_flags);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev1, e->stream));
    if (!s.inflated) HIP_TRY(hipEventCreateWithFlags(&s.inflated, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(s.inflated, e->stream));
    e->inflate_events.emplace_back(ev0, ev1);
    e->inflated_bytes += data_end - FEED_GAP;
    {   // how full the launch was: a block per lane, the kernel's grid is n_cu x INF_WAVES_PER_CU waves, a partial round costs a whole one
        const uint64_t lanes = (uint64_t)e->n_cu * (uint64_t)pssbam::INF_WAVES_PER_CU * 64ull;
        e->feed_blocks_launched += nb;
        e->feed_lanes_launched += (nb + lanes - 1) / lanes * lanes;
    }
    // the tally: now, or -- inflated ahead of the genome -- when set_references comes
    const bool ready = feed_engine_ready(e);
    for (size_t k = 0; k < s.sub_first.size(); k++) {
        DeferredTally d;
        d.slot = e->cur_feed;
        d.sub_base = sub_base[k];
        d.sub_len = sub_len[k];
        d.offs_at = sub_offs[k];
        d.n_bound = (uint32_t)std::min<uint64_t>((k + 1 < sub_offs.size() ? sub_offs[k + 1] : offs_at) - sub_offs[k], 0xFFFFFFF0ull);
        d.k = (uint32_t)k;
        d.sample_off = k == 0 ? FEED_GAP : 0ull;
        d.last_of_slot = k + 1 == s.sub_first.size();
        s.held = true;
        if (ready) { if ((rc = feed_launch_tally(e, d))) return rc; }
        else e->deferred.push_back(d);
    }
    s.busy = true;
    s.flush_seq = ++e->flush_seq;
    s.blocks.clear();
    s.sub_first.clear();
    s.comp_used = s.out_used = s.sub_bytes = 0;
    e->cur_feed = -1;
    return PSSBAM_OK;
}

// appends blocks[b0, b1) of a chunk (and their compressed bytes) to the super-batch being assembled
static int feed_append(pssbam_engine *e, const uint8_t *comp, const pssbam_bgzf_block *blocks, uint32_t b0, uint32_t b1, hipStream_t cs) {
    FeedAcc &s = *e->feed[(size_t)e->cur_feed];
    if (s.blocks.empty()) s.out_used = FEED_GAP;
    const uint64_t byte0 = blocks[b0].in_off & ~15ull, byte1 = blocks[b1 - 1].in_off + blocks[b1 - 1].in_len;
    const uint64_t out_bytes = blocks[b1 - 1].out_off + blocks[b1 - 1].isize - blocks[b0].out_off;
    if (s.out_cap < s.out_used + out_bytes + 8192) return fail(PSSBAM_ESTATE, "output buffer of the super-batch is too small");
    // blocks: in_off -> into d_comp, out_off -> into d_out, back to back (records may cross blocks); a new
    // tally sub-batch where the record bytes would pass 3.5 GiB
    const uint64_t comp_at = (s.comp_used + 15ull) & ~15ull;
    if (comp_at + (byte1 - byte0) + 32 > s.comp_cap) return fail(PSSBAM_ESTATE, "compressed bytes of the super-batch exceed their buffer");
    for (uint32_t i = b0; i < b1; i++) {
        pssbam_bgzf_block b = blocks[i];
        if (s.sub_first.empty() || s.sub_bytes + b.isize > FEED_SUB_MAX) {
            s.sub_first.push_back((uint32_t)s.blocks.size());
            s.sub_bytes = 0;
        }
        b.in_off = b.in_off - byte0 + comp_at;
        b.out_off = s.out_used;
        b.status = 0;
        s.out_used += b.isize;
        s.sub_bytes += b.isize;
        s.blocks.push_back(b);
    }
    HIP_TRY(hipMemcpyAsync(s.d_comp + comp_at, comp + byte0, byte1 - byte0, hipMemcpyHostToDevice, cs));
    s.comp_used = comp_at + (byte1 - byte0);
    e->h2d_bytes += byte1 - byte0;
    return PSSBAM_OK;
}

// the feed's per-engine device words and targets, made at the first use of the feed
static int feed_state_init(pssbam_engine *e) {
    if (e->d_feed_flags) return PSSBAM_OK;
    e->feed_t0 = feed_now();
    if (getenv("PSSBAM_FEED_SUPER_BYTES")) e->feed_out_target = std::max<uint64_t>(1ull << 20, strtoull(getenv("PSSBAM_FEED_SUPER_BYTES"), nullptr, 10));
    // the inflate kernel keeps INF_WAVES_PER_CU waves x 64 lanes per CU busy, a block per lane, and blocks take about the
    // same time: a super-batch of a whole number of "rounds" of blocks wastes no partial round
    const uint64_t lanes = (uint64_t)e->n_cu * (uint64_t)pssbam::INF_WAVES_PER_CU * 64ull;
    const uint64_t rounds = std::max<uint64_t>(1, e->feed_out_target / (lanes * 65280ull));
    e->feed_block_target = getenv("PSSBAM_FEED_SUPER_BYTES") && e->feed_out_target < lanes * 65280ull ? 0xFFFFFFFFull : rounds * lanes;
    HIP_TRY(hipMalloc(&e->d_feed_flags, sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(e->d_feed_flags, 0, sizeof(uint32_t), e->stream));
    HIP_TRY(hipMalloc(&e->d_feed_tail, sizeof(uint64_t)));
    HIP_TRY(hipMemsetAsync(e->d_feed_tail, 0, sizeof(uint64_t), e->stream));
    if (!e->d_carry) HIP_TRY(hipMalloc(&e->d_carry, FEED_GAP));
    if (const char *ib = getenv("PSSBAM_FEED_IDLE_BLOCKS")) e->feed_idle_min_blocks = atoi(ib) > 0 ? (uint64_t)atoi(ib) : ~0ull;
    return PSSBAM_OK;
}

// Declares that compressed blocks will be fed BEFORE the genome is set: n_ref = the reference count of the BAM
// header (the record chain is judged with it).  submit_bgzf is then legal at once; inflate, CRC-32 and the record
// index run as the blocks arrive, the tally launches follow when set_genome(_async) + set_references have been
// called.  genome_bytes_hint (0 = unknown): device memory to leave alone for the genome, e.g. the FASTA's size.
extern "C" int pssbam_engine_feed_open(pssbam_engine *e, int32_t n_ref, uint64_t genome_bytes_hint) {
    if (!e || n_ref < 0) return fail(PSSBAM_EINVAL, "bad argument");
    if (e->cfg.kernel == PSSBAM_KERNEL_SIMPLE) return fail(PSSBAM_EINVAL, "device-indexed blocks need the tiled kernels");
    HIP_TRY(hipSetDevice(e->device));
    e->feed_opened = true;
    e->feed_n_ref = n_ref;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const uint64_t keep = genome_bytes_hint + genome_bytes_hint / 2 + (8ull << 30);   // genome + 4-bit image + slack
        e->feed_mem_budget = e->feed.size() * feed_slot_bytes(e) + (free_b > keep ? free_b - keep : 0);
    }
    return PSSBAM_OK;
}

extern "C" int pssbam_engine_submit_bgzf(pssbam_engine *e, const void *comp, uint64_t comp_bytes, const pssbam_bgzf_block *blocks,
                                         uint32_t n_blocks, uint32_t first_record_offset, uint64_t *ticket) {
    if (!e) return fail(PSSBAM_EINVAL, "null engine");
    if (!feed_engine_ready(e) && !e->feed_opened) {
        int rc = check_ready(e);
        if (rc) return rc;
    }
    int rc;
    if (ticket) *ticket = 0;
    if (!n_blocks) return PSSBAM_OK;
    if (!comp || !blocks) return fail(PSSBAM_EINVAL, "null buffer");
    if (e->cfg.kernel == PSSBAM_KERNEL_SIMPLE) return fail(PSSBAM_EINVAL, "device-indexed blocks need the tiled kernels");
    const uint64_t out_bytes = blocks[n_blocks - 1].out_off + blocks[n_blocks - 1].isize;
    if (out_bytes > (1ull << 30)) return fail(PSSBAM_EINVAL, "chunk inflates to %llu bytes; keep chunks at or below 1 GiB", (unsigned long long)out_bytes);
    if (blocks[0].out_off != 0) return fail(PSSBAM_EINVAL, "blocks[0].out_off must be 0");
    if (first_record_offset > blocks[0].isize) return fail(PSSBAM_EINVAL, "first_record_offset lies beyond the first block");
    // the kernels trust this table for every address they form: check it here, once, on the host
    for (uint32_t i = 0; i < n_blocks; i++) {
        const pssbam_bgzf_block &b = blocks[i];
        if (b.isize > 65536u || b.in_len > 65536u) return fail(PSSBAM_EINVAL, "blocks[%u]: a BGZF block holds at most 64 KiB (in_len %u, isize %u)", i, b.in_len, b.isize);
        if (b.in_off > comp_bytes || b.in_len > comp_bytes - b.in_off) return fail(PSSBAM_EINVAL, "blocks[%u]: payload lies outside the %llu-byte chunk", i, (unsigned long long)comp_bytes);
        if (i && (b.in_off < blocks[i - 1].in_off + blocks[i - 1].in_len || b.out_off != blocks[i - 1].out_off + blocks[i - 1].isize))
            return fail(PSSBAM_EINVAL, "blocks[%u]: blocks must be in file order with contiguous out_off", i);
    }
    const bool assembling = e->cur_feed >= 0 && !e->feed[(size_t)e->cur_feed]->blocks.empty();
    if (first_record_offset && !(e->feed_fresh && !assembling))
        return fail(PSSBAM_ESTATE, "first_record_offset only makes sense for the first blocks of a stream (after create / reset)");
    if (first_record_offset) e->feed_skip = first_record_offset;
    if (comp_bytes > e->feed_comp_cap / 2) return fail(PSSBAM_EINVAL, "chunk of %llu compressed bytes is too large", (unsigned long long)comp_bytes);
    HIP_TRY(hipSetDevice(e->device));
    if ((rc = feed_state_init(e))) return rc;
    // All or nothing: while the tallies are put off a slot may be unobtainable (PSSBAM_EBUSY) -- find that out before
    // the first block of this chunk is taken.  A chunk spills over into at most one more super-batch (<= 1 GiB against
    // 4.3 GB); with tiny test super-batches it may need more, which then go past the budget rather than fail half-way.
    if (!feed_engine_ready(e)) {
        bool fits = false;
        if (e->cur_feed >= 0) {
            const FeedAcc &cur = *e->feed[(size_t)e->cur_feed];
            const uint64_t used = std::max<uint64_t>(cur.out_used, FEED_GAP);
            fits = cur.blocks.size() + n_blocks < e->feed_block_target && used + out_bytes + 8192 <= FEED_GAP + e->feed_out_target &&
                   cur.comp_used + comp_bytes + 64 <= e->feed_comp_cap;
        }
        if (!fits && e->spare_feed < 0) {
            rc = feed_acquire(e, &e->spare_feed, false);
            if (rc) return rc;
        }
    }
    late_streams(e, false);
    hipStream_t cs = (e->ticket_seq & 1u) && e->copy_stream2 ? e->copy_stream2 : e->copy_stream ? e->copy_stream : e->stream;
    uint32_t b0 = 0;
    while (b0 < n_blocks) {
        if (e->cur_feed < 0) {
            if (e->spare_feed >= 0) { e->cur_feed = e->spare_feed; e->spare_feed = -1; }
            else if ((rc = feed_acquire(e, &e->cur_feed, true))) { e->cur_feed = -1; return rc; }
            FeedAcc &n = *e->feed[(size_t)e->cur_feed];
            n.blocks.clear();
            n.sub_first.clear();
            n.comp_used = n.sub_bytes = 0;
            n.out_used = FEED_GAP;
        }
        FeedAcc &cur = *e->feed[(size_t)e->cur_feed];
        // (the first super-batch of a stream is cut at half a round of blocks / a third of the bytes, so the device starts earlier)
        const bool first = e->flush_seq == 0;
        const uint64_t byte_target = first ? e->feed_out_target / 3 : e->feed_out_target;
        const uint64_t block_target = e->feed_block_target == 0xFFFFFFFFull ? 0xFFFFFFFFull
                                      : first ? (uint64_t)e->n_cu * 64ull * pssbam::INF_WAVES_PER_CU / 2ull   // half a round: the device starts early
                                              : e->feed_block_target;
        // how many of the remaining blocks still fit
        uint32_t b1 = b0;
        const uint64_t comp0 = blocks[b0].in_off & ~15ull, out0 = blocks[b0].out_off;
        while (b1 < n_blocks && cur.blocks.size() + (b1 - b0) < block_target &&
               cur.out_used - FEED_GAP + (blocks[b1].out_off + blocks[b1].isize - out0) <= byte_target + FEED_OVERSHOOT &&
               cur.out_used + (blocks[b1].out_off + blocks[b1].isize - out0) + 8192 <= cur.out_cap &&
               cur.comp_used + (blocks[b1].in_off + blocks[b1].in_len - comp0) + 64 <= e->feed_comp_cap)
            b1++;
        if (b1 == b0) {
            if (cur.blocks.empty()) return fail(PSSBAM_EINVAL, "a single BGZF block does not fit the feed buffers");
            rc = feed_flush(e);
            if (rc) return rc;
            continue;
        }
        rc = feed_append(e, (const uint8_t *)comp, blocks, b0, b1, cs);
        if (rc) return rc;
        b0 = b1;
        bool flush_now = cur.blocks.size() >= block_target || cur.out_used - FEED_GAP >= byte_target;
        if (!flush_now && cur.blocks.size() >= e->feed_idle_min_blocks && e->flush_seq > 0) {
            // The device has run dry (everything flushed so far has been inflated) while this super-batch is still being
            // collected -- the loaders are the slower side, typically while the caller's FASTA parser has the CPUs: a partial
            // round now (the lanes without a block idle) beats a full one later
            const FeedAcc *last = nullptr;
            for (const FeedAcc *sp : e->feed)
                if (sp->busy && sp->flush_seq == e->flush_seq) last = sp;
            if (last && hipEventQuery(last->copies_done) == hipSuccess && hipEventQuery(last->inflated) == hipSuccess) { flush_now = true; e->feed_early_flushes++; }
            (void)hipGetLastError();
        }
        if (flush_now) {
            rc = feed_flush(e);
            if (rc) return rc;
        }
    }
    hipEvent_t done = nullptr;
    if (!e->feed_event_pool.empty()) { done = e->feed_event_pool.back(); e->feed_event_pool.pop_back(); }
    else HIP_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(done, cs));
    const uint64_t t = ++e->ticket_seq;
    e->feed_copies.emplace_back(t, done);
    if (ticket) *ticket = t;
    return PSSBAM_OK;
}

// The blocks submitted next do not continue the stream fed so far (e.g. this engine is dealt every
// n-th run of a file): what is pending is flushed, a partial record left at this point is an error
// (PSSBAM_FEED_TRUNCATED -- it can never be completed), and the next blocks start a new chain at a
// record boundary.
extern "C" int pssbam_engine_feed_break(pssbam_engine *e) {
    if (!e) return fail(PSSBAM_EINVAL, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    int rc = feed_flush(e);
    if (rc) return rc;
    if (e->d_feed_tail) {
        hipLaunchKernelGGL(pssbam::bgzf_chain_break, dim3(1), dim3(64), 0, e->stream, e->d_feed_tail, e->d_feed_flags);
        HIP_TRY(hipGetLastError());
    }
    e->feed_fresh = true;
    e->feed_skip = 0;
    return PSSBAM_OK;
}

// Several engines are dealt alternating runs of ONE stream (one BAM, n GPUs): the blocks submitted to `to` from now on
// continue the stream where the blocks submitted to `from` so far end.  What `from` has pending is flushed; the partial
// record its run ends with (records cross BGZF blocks in files written by htsjdk) travels to `to` -- a few hundred bytes
// through page-locked host memory, device to device by two small kernels and an event, no host wait -- and is completed,
// indexed and tallied there.  The record chain is thus checked across the engines exactly as inside one.
extern "C" int pssbam_engine_feed_handoff(pssbam_engine *from, pssbam_engine *to) {
    if (!from || !to) return fail(PSSBAM_EINVAL, "null engine");
    if (from == to) return PSSBAM_OK;
    if (to->cfg.kernel == PSSBAM_KERNEL_SIMPLE) return fail(PSSBAM_EINVAL, "device-indexed blocks need the tiled kernels");
    if (to->cur_feed >= 0 && !to->feed[(size_t)to->cur_feed]->blocks.empty())
        return fail(PSSBAM_ESTATE, "the receiving engine is in the middle of a run (hand its own run on, or break it, first)");
    if (!from->d_feed_tail) return fail(PSSBAM_ESTATE, "the engine handing on has been fed nothing");
    int rc;
    HIP_TRY(hipSetDevice(to->device));
    if ((rc = feed_state_init(to))) return rc;
    if (!to->h_handoff) HIP_TRY(hipHostMalloc((void **)&to->h_handoff, 8 + FEED_GAP, hipHostMallocPortable));
    HIP_TRY(hipSetDevice(from->device));
    if ((rc = feed_flush(from))) return rc;
    if (!from->handoff_ev) HIP_TRY(hipEventCreateWithFlags(&from->handoff_ev, hipEventDisableTiming));
    hipLaunchKernelGGL(pssbam::bgzf_chain_handoff_out, dim3(1), dim3(256), 0, from->stream, (const uint8_t *)from->d_carry, from->d_feed_tail,
                       to->h_handoff + 8, (uint64_t *)to->h_handoff);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(from->handoff_ev, from->stream));
    from->feed_fresh = false;   // (its next run, if any, starts with what is handed to IT)
    HIP_TRY(hipSetDevice(to->device));
    HIP_TRY(hipStreamWaitEvent(to->stream, from->handoff_ev, 0));
    hipLaunchKernelGGL(pssbam::bgzf_chain_handoff_in, dim3(1), dim3(256), 0, to->stream, (const uint8_t *)(to->h_handoff + 8), (const uint64_t *)to->h_handoff,
                       to->d_carry, (uint64_t)FEED_GAP, to->d_feed_tail);
    HIP_TRY(hipGetLastError());
    to->feed_fresh = false;
    to->feed_skip = 0;
    return PSSBAM_OK;
}

// copy completion of a submit_bgzf ticket (the caller's compressed chunk is then free)
extern "C" int pssbam_engine_wait_bgzf_copied(pssbam_engine *e, uint64_t ticket) {
    if (!e) return fail(PSSBAM_EINVAL, "null engine");
    if (!ticket) return PSSBAM_OK;
    for (size_t i = 0; i < e->feed_copies.size(); i++)
        if (e->feed_copies[i].first == ticket) {
            HIP_TRY(hipSetDevice(e->device));
            HIP_TRY(hipEventSynchronize(e->feed_copies[i].second));
            e->feed_event_pool.push_back(e->feed_copies[i].second);
            e->feed_copies.erase(e->feed_copies.begin() + (long)i);
            return PSSBAM_OK;
        }
    return PSSBAM_OK;
}

// Drains the engine and reports what the device-side feed saw: *flags = OR of 1 (a block failed
// inflate / ISIZE / CRC-32), 2 (records cross BGZF blocks: the device index cannot be used, fall
// back to the host reader), 4 (a record length below 32); inflate_ms / inflated_bytes = summed
// inflate + CRC + index kernel time and payload.
extern "C" int pssbam_engine_feed_status(pssbam_engine *e, uint32_t *flags, double *inflate_ms, uint64_t *inflated_bytes) {
    if (!e) return fail(PSSBAM_EINVAL, "null engine");
    int rc = pssbam_engine_sync(e);
    if (rc) return rc;
    uint32_t f = 0;
    if (e->d_feed_flags) HIP_TRY(hipMemcpy(&f, e->d_feed_flags, sizeof f, hipMemcpyDeviceToHost));
    if (e->d_feed_tail) {   // a record cut off by the end of the stream
        uint64_t tail = 0;
        HIP_TRY(hipMemcpy(&tail, e->d_feed_tail, sizeof tail, hipMemcpyDeviceToHost));
        if (tail) f |= pssbam::FEED_TRUNCATED;
    }
    {   // kernel time of the feed = the UNION of the super-batches' intervals (first inflate instruction .. record index written):
        // consecutive launches overlap on purpose, a plain sum would count the overlap twice
        std::vector<std::pair<float, float>> iv;
        for (auto &p : e->inflate_events) {
            float a = 0.f, b = 0.f;
            HIP_TRY(hipEventElapsedTime(&a, e->feed_base_ev, p.first));
            HIP_TRY(hipEventElapsedTime(&b, e->feed_base_ev, p.second));
            iv.emplace_back(a, b);
            e->event_pool.push_back(p.first);
            e->event_pool.push_back(p.second);
        }
        e->inflate_events.clear();
        std::sort(iv.begin(), iv.end());
        float hi = -1e30f;
        for (auto &x : iv) {
            if (x.second <= hi) continue;
            e->inflate_ms += x.second - std::max(x.first, hi);
            hi = x.second;
        }
        if (e->feed_base_ev) { (void)hipEventDestroy(e->feed_base_ev); e->feed_base_ev = nullptr; }
    }
    if (getenv("PSSBAM_STATS"))
        fprintf(stderr, "[pssbam] engine feed: %llu super-batches (%llu cut short because the device had run dry; their blocks filled %.0f %% of the lanes of the inflate launches) in %zu slots (%llu allocated here), %llu tally launches put off until the genome was set; "
                        "buffer allocation %.3f, waiting for a busy slot %.3f, flush (block table + launches) %.3f s; first launch %.3f s after the first block came\n",
                (unsigned long long)e->flush_seq, (unsigned long long)e->feed_early_flushes,
                e->feed_lanes_launched ? 100.0 * (double)e->feed_blocks_launched / (double)e->feed_lanes_launched : 0.0, e->feed.size(), (unsigned long long)e->feed_slots_allocated, (unsigned long long)e->feed_deferred_launches,
                e->feed_t_alloc, e->feed_t_wait_busy, e->feed_t_flush, e->feed_t_first_flush);
    if (flags) *flags = f;
    if (inflate_ms) *inflate_ms = e->inflate_ms;
    if (inflated_bytes) *inflated_bytes = e->inflated_bytes;
    return PSSBAM_OK;
}
