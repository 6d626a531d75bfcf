// pss-bam_amd/csrc/bgzf_api.h -- C ABI of the device-side BGZF inflate (included by engine.hip).
#pragma once

#include "inflate_kernels.h"

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
}  // namespace

static_assert(sizeof(pssbam_bgzf_block) == sizeof(pssbam::BgzfBlock), "public and device block descriptors must match");

extern "C" int pssbam_bgzf_inflate_device(void *hip_stream, const void *d_comp, uint64_t comp_bytes, pssbam_bgzf_block *d_blocks,
                                          uint32_t n_blocks, void *d_out, int check_crc) {
    if (!n_blocks) return PSSBAM_OK;
    if (!d_comp || !d_blocks || !d_out) return fail(PSSBAM_EINVAL, "null buffer");
    if ((uintptr_t)d_comp & 3u) return fail(PSSBAM_EINVAL, "d_comp must be 4-byte aligned");
    int dev = 0, n_cu = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    hipStream_t st = (hipStream_t)hip_stream;
    static bool attr_set[64] = {false};
    if (!attr_set[dev & 63]) {
        HIP_TRY(hipFuncSetAttribute((const void *)pssbam::bgzf_inflate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pssbam::INF_LDS_BYTES));
        attr_set[dev & 63] = true;
    }
    const uint32_t groups = (n_blocks + pssbam::INF_WAVE - 1) / pssbam::INF_WAVE;
    const char *gm = getenv("PSSBAM_INFLATE_WAVES_PER_CU");
    const uint32_t per_cu = gm && atoi(gm) > 0 ? (uint32_t)atoi(gm) : 3u;
    const uint32_t grid = std::min<uint32_t>(groups, (uint32_t)n_cu * per_cu);
    hipLaunchKernelGGL(pssbam::bgzf_inflate_kernel, dim3(grid), dim3(pssbam::INF_WAVE), pssbam::INF_LDS_BYTES, st, (const uint8_t *)d_comp, comp_bytes,
                       (pssbam::BgzfBlock *)d_blocks, n_blocks, (uint8_t *)d_out);
    HIP_TRY(hipGetLastError());
    if (check_crc) {
        uint32_t *xpow = nullptr;
        int rc = ensure_xpow(dev, &xpow);
        if (rc) return rc;
        const uint32_t cgrid = std::min<uint32_t>((n_blocks + 3) / 4, (uint32_t)n_cu * 8u);
        hipLaunchKernelGGL(pssbam::bgzf_crc_kernel, dim3(cgrid), dim3(256), 0, st, (const uint8_t *)d_out, (pssbam::BgzfBlock *)d_blocks, n_blocks, xpow);
        HIP_TRY(hipGetLastError());
    }
    return PSSBAM_OK;
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
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = PSSBAM_OK;
    auto cleanup = [&]() {
        if (d_comp) (void)hipFree(d_comp);
        if (d_out) (void)hipFree(d_out);
        if (d_blocks) (void)hipFree(d_blocks);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
#define TRY_C(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return fail(PSSBAM_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e)); } } while (0)
    TRY_C(hipMalloc(&d_comp, nbytes + 16));
    TRY_C(hipMalloc(&d_out, total + 16));
    TRY_C(hipMalloc(&d_blocks, (size_t)n * sizeof(pssbam_bgzf_block)));
    TRY_C(hipMemset(d_comp + nbytes, 0, 16));
    TRY_C(hipMemcpy(d_comp, bgzf, nbytes, hipMemcpyHostToDevice));
    TRY_C(hipMemcpy(d_blocks, blocks.data(), (size_t)n * sizeof(pssbam_bgzf_block), hipMemcpyHostToDevice));
    TRY_C(hipEventCreate(&e0));
    TRY_C(hipEventCreate(&e1));
    if (repeats < 1) repeats = 1;
    float best = 1e30f;
    for (int r = 0; r < repeats && rc == PSSBAM_OK; r++) {
        TRY_C(hipEventRecord(e0, nullptr));
        rc = pssbam_bgzf_inflate_device(nullptr, d_comp, nbytes, d_blocks, (uint32_t)n, d_out, check_crc);
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
// the feed: compressed batch -> device inflate -> device record index -> tally, one call
// --------------------------------------------------------------------------------------
template <class T>
static int grow(T **ptr, size_t *cap, size_t need, size_t elem = sizeof(T)) {
    if (*cap >= need) return PSSBAM_OK;
    if (*ptr) HIP_TRY(hipFree(*ptr));
    *ptr = nullptr;
    *cap = need + need / 8 + 4096;
    HIP_TRY(hipMalloc((void **)ptr, *cap * elem));
    return PSSBAM_OK;
}

extern "C" int pssbam_engine_submit_bgzf(pssbam_engine *e, const void *comp, uint64_t comp_bytes, const pssbam_bgzf_block *blocks,
                                         uint32_t n_blocks, uint32_t first_record_offset, uint64_t *ticket) {
    int rc = check_ready(e);
    if (rc) return rc;
    if (ticket) *ticket = 0;
    if (!n_blocks) return PSSBAM_OK;
    if (!comp || !blocks) return fail(PSSBAM_EINVAL, "null buffer");
    const uint64_t out_bytes = blocks[n_blocks - 1].out_off + blocks[n_blocks - 1].isize;
    if (out_bytes >= (1ull << 32) - (1ull << 16)) return fail(PSSBAM_EINVAL, "batch inflates to %llu bytes; keep batches below 4 GiB", (unsigned long long)out_bytes);
    if (blocks[0].out_off != 0) return fail(PSSBAM_EINVAL, "blocks[0].out_off must be 0");
    if (first_record_offset > blocks[0].isize) return fail(PSSBAM_EINVAL, "first_record_offset lies beyond the first block");
    HIP_TRY(hipSetDevice(e->device));
    if (!e->d_feed_flags) {
        HIP_TRY(hipMalloc(&e->d_feed_flags, sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(e->d_feed_flags, 0, sizeof(uint32_t), e->stream));
    }
    FeedSlot &s = e->feed[e->next_feed];
    e->next_feed ^= 1;
    if (!s.copied) {
        HIP_TRY(hipEventCreate(&s.copy_begin));
        HIP_TRY(hipEventCreate(&s.copied));
        HIP_TRY(hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming));
    }
    if (s.busy) HIP_TRY(hipEventSynchronize(s.consumed));
    s.busy = false;
    if (s.timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.copy_begin, s.copied) == hipSuccess) e->h2d_ms += ms;
        s.timed = false;
    }
    const uint64_t max_recs = out_bytes / 36ull + 2ull;   // a record is at least 36 bytes
    size_t bc = s.blocks_cap, bc2 = s.blocks_cap, bc3 = s.blocks_cap;
    if ((rc = grow(&s.d_comp, &s.comp_cap, (size_t)comp_bytes + 16))) return rc;
    if ((rc = grow((uint8_t **)&s.d_blocks, &bc, (size_t)n_blocks, sizeof(pssbam::BgzfBlock)))) return rc;
    if ((rc = grow(&s.d_counts, &bc2, (size_t)n_blocks))) return rc;
    if ((rc = grow(&s.d_base, &bc3, (size_t)n_blocks))) return rc;
    s.blocks_cap = std::min(bc, std::min(bc2, bc3));
    if ((rc = grow(&s.d_out, &s.out_cap, (size_t)out_bytes + 64))) return rc;
    if ((rc = grow(&s.d_offs, &s.offs_cap, (size_t)max_recs))) return rc;
    if (!s.d_nrecs) HIP_TRY(hipMalloc(&s.d_nrecs, sizeof(uint32_t)));

    // compressed bytes + block table over PCIe (two copy streams for large batches, like submit)
    HIP_TRY(hipEventRecord(s.copy_begin, e->copy_stream));
    const uint64_t half = comp_bytes >= (32ull << 20) ? (comp_bytes / 2) & ~4095ull : 0;
    if (half) {
        HIP_TRY(hipStreamWaitEvent(e->copy_stream2, s.copy_begin, 0));
        HIP_TRY(hipMemcpyAsync(s.d_comp + half, (const uint8_t *)comp + half, comp_bytes - half, hipMemcpyHostToDevice, e->copy_stream2));
        HIP_TRY(hipEventRecord(e->copied2, e->copy_stream2));
    }
    HIP_TRY(hipMemcpyAsync(s.d_comp, comp, half ? half : comp_bytes, hipMemcpyHostToDevice, e->copy_stream));
    HIP_TRY(hipMemcpyAsync(s.d_blocks, blocks, (size_t)n_blocks * sizeof(pssbam_bgzf_block), hipMemcpyHostToDevice, e->copy_stream));
    if (half) HIP_TRY(hipStreamWaitEvent(e->copy_stream, e->copied2, 0));
    HIP_TRY(hipEventRecord(s.copied, e->copy_stream));
    s.timed = true;
    e->h2d_bytes += comp_bytes + (uint64_t)n_blocks * sizeof(pssbam_bgzf_block);
    HIP_TRY(hipStreamWaitEvent(e->stream, s.copied, 0));

    // inflate + CRC + record index on the engine's stream
    hipEvent_t ev0 = take_event(e), ev1 = take_event(e);
    if (!ev0 || !ev1) return fail(PSSBAM_EHIP, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(ev0, e->stream));
    rc = pssbam_bgzf_inflate_device(e->stream, s.d_comp, comp_bytes, (pssbam_bgzf_block *)s.d_blocks, n_blocks, s.d_out,
                                    getenv("PSSBAM_NO_CRC") ? 0 : 1);
    if (rc) return rc;
    const uint32_t igrid = std::min<uint32_t>((n_blocks + 255u) / 256u, (uint32_t)e->n_cu * 8u);
    hipLaunchKernelGGL(pssbam::bgzf_index_count, dim3(igrid), dim3(256), 0, e->stream, (const uint8_t *)s.d_out,
                       (const pssbam::BgzfBlock *)s.d_blocks, n_blocks, first_record_offset, s.d_counts, e->d_feed_flags);
    hipLaunchKernelGGL(pssbam::bgzf_index_scan, dim3(1), dim3(1024), 0, e->stream, (const uint32_t *)s.d_counts, n_blocks, s.d_base, s.d_nrecs);
    hipLaunchKernelGGL(pssbam::bgzf_index_write, dim3(igrid), dim3(256), 0, e->stream, (const uint8_t *)s.d_out,
                       (const pssbam::BgzfBlock *)s.d_blocks, n_blocks, first_record_offset, (const uint32_t *)s.d_counts,
                       (const uint32_t *)s.d_base, s.d_offs, (const uint32_t *)s.d_nrecs, (uint32_t)out_bytes);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev1, e->stream));
    e->inflate_events.emplace_back(ev0, ev1);
    e->inflated_bytes += out_bytes;

    rc = launch_tally(e, s.d_out, out_bytes, s.d_offs, (uint32_t)std::min<uint64_t>(max_recs, 0xFFFFFFF0ull), nullptr, 0, s.d_nrecs,
                      first_record_offset);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(s.consumed, e->stream));
    s.busy = true;
    s.ticket = ++e->ticket_seq;
    if (ticket) *ticket = s.ticket;
    return PSSBAM_OK;
}

// copy completion of a submit_bgzf ticket (the compressed staging buffer is then free)
extern "C" int pssbam_engine_wait_bgzf_copied(pssbam_engine *e, uint64_t ticket) {
    if (!e) return fail(PSSBAM_EINVAL, "null engine");
    if (!ticket) return PSSBAM_OK;
    for (FeedSlot &s : e->feed)
        if (s.ticket == ticket) {
            HIP_TRY(hipSetDevice(e->device));
            HIP_TRY(hipEventSynchronize(s.copied));
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
    for (auto &p : e->inflate_events) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
        e->inflate_ms += ms;
        e->event_pool.push_back(p.first);
        e->event_pool.push_back(p.second);
    }
    e->inflate_events.clear();
    if (flags) *flags = f;
    if (inflate_ms) *inflate_ms = e->inflate_ms;
    if (inflated_bytes) *inflated_bytes = e->inflated_bytes;
    return PSSBAM_OK;
}
