/* pss-bam_amd/host/device_feed.h -- BAM feed with the BGZF inflate on the GPU (device_feed.c). */
#ifndef PSSBAM_DEVICE_FEED_H
#define PSSBAM_DEVICE_FEED_H

#include <stddef.h>
#include <stdint.h>

#include "pssbam_hip.h"

typedef struct device_feed_stats {
    uint64_t n_submits, compressed_bytes, inflated_bytes;
    double inflate_ms;   /* summed inflate + CRC + index kernel time over the engines */
    uint32_t flags;      /* PSSBAM_FEED_* seen */
    int fallback;        /* 1: the file must go through the host reader instead (records cross BGZF
                            blocks, or the input is not a regular file); the engines then hold
                            partial counts and must be reset */
} device_feed_stats;

/* $PSSBAM_DEVICE_INFLATE (default 1) */
int device_feed_enabled(void);

/* The feed may start BEFORE the genome is on the engines (pssbam_engine_feed_open): inflate, CRC and record
 * index need no reference base.  The caller then hands in a gate: poll(ctx, block) returns 1 once genome and
 * references are set on every engine -- setting them itself, on the calling thread, the first time it finds
 * the genome available; 0 = not there yet (block == 0 only); < 0 = give up.  The feed polls it between
 * submits, blocks on it when an engine answers PSSBAM_EBUSY, and in any case before it drains the engines. */
typedef struct feed_gate {
    int (*poll)(void *ctx, int block);
    void *ctx;
} feed_gate;

/* Streams the BGZF file at `path` through eng[0..n_gpus) (genome and references already set):
 * compressed chunks over PCIe, inflate + CRC + record index + tally on the devices, runs of `run`
 * consecutive batches per engine.  header_bytes = inflated bytes in front of the first alignment
 * record.  0 = done (see fs->fallback), -1 = failed after printing a diagnostic. */
int run_device_feed(pssbam_engine *const *eng, int n_gpus, const char *path, size_t header_bytes, int run, int verbose,
                    device_feed_stats *fs, const feed_gate *gate /* NULL: the engines are ready */);

/* Start-up overlap for the front ends: opens the file and starts the loader threads at once,
 * so the first windows sit in the staging slots by the time the genome is on the device;
 * run_device_feed() on the same path takes the loader over.  _pin page-locks the slots (needs the HIP
 * runtime up), _cancel drops a loader nobody took. */
void device_feed_prefetch(const char *path);
void device_feed_prefetch_pin(void);
void device_feed_prefetch_cancel(void);
#endif
