/*
 * pss-bam_amd/host/hostcheck_main.c -- `hostcheck [-f genome.fa] [-a alignments]`: drives every
 * multi-threaded piece of the host library without a GPU and prints a digest of what came out:
 * the FASTA loader (block + parallel parser), the three-stage BGZF/BAM reader and the threaded
 * SAM-text reader.  It exists for the sanitizer builds (`make sanitize`: ThreadSanitizer and
 * AddressSanitizer + UBSan; tests/test_sanitizers.py) -- the digests of the instrumented
 * binaries must equal the plain build's, and the sanitizers must stay silent.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bam_reader.h"
#include "fasta-genome-io.h"
#include "sam_reader.h"

static uint64_t fnv(uint64_t h, const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

int main(int argc, char **argv)
{
    const char *fa = NULL, *aln = NULL;
    int quick = 0; /* -q: no digest, just drain the reader and report its inflate seconds (throughput probe) */
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-q")) { quick = 1; continue; }
        if (!strcmp(argv[i], "-f") && i + 1 < argc) fa = argv[++i];
        else if (!strcmp(argv[i], "-a") && i + 1 < argc) aln = argv[++i];
    }
    if (!fa && !aln) {
        fprintf(stderr, "usage: hostcheck [-f genome.fa[.gz]] [-a file.bam|file.sam[.gz]]\n");
        return 2;
    }
    if (fa) {
        Genome *g = init_genome(fa);
        if (!g) { fprintf(stderr, "hostcheck: cannot load %s\n", fa); return 1; }
        uint64_t h = 1469598103934665603ull, bases = 0;
        for (size_t i = 0; i < g->n_seqs; i++) {
            h = fnv(h, g->seqs[i]->id, strlen(g->seqs[i]->id) + 1);
            h = fnv(h, g->seqs[i]->seq, g->seqs[i]->len);
            bases += g->seqs[i]->len;
        }
        printf("genome seqs=%zu bases=%llu digest=%016llx\n", (size_t)g->n_seqs, (unsigned long long)bases, (unsigned long long)h);
        destroy_genome(g);
    }
    if (aln) {
        char err[512];
        const int is_bam = file_is_bam(aln);
        if (is_bam < 0) { fprintf(stderr, "hostcheck: cannot open %s\n", aln); return 1; }
        bam_reader *rd = is_bam ? bam_reader_open(aln, 0, 0, err, sizeof err) : NULL;
        sam_reader *sd = is_bam ? NULL : sam_reader_open(aln, 0, err, sizeof err);
        if (!rd && !sd) { fprintf(stderr, "hostcheck: %s\n", err); return 1; }
        uint64_t h = 1469598103934665603ull, recs = 0, bytes = 0, batches = 0;
        for (;;) {
            const uint8_t *p;
            const uint32_t *offs;
            size_t nbytes;
            const int64_t n = rd ? bam_reader_next(rd, &p, &offs, &nbytes) : sam_reader_next(sd, &p, &offs, &nbytes);
            if (n < 0) { fprintf(stderr, "hostcheck: %s\n", rd ? bam_reader_error(rd) : sam_reader_error(sd)); return 1; }
            if (n == 0) break;
            if (offs[0] != 0 || offs[n] != nbytes) { fprintf(stderr, "hostcheck: offset index does not cover the batch\n"); return 1; }
            for (int64_t i = 0; i < n && !quick; i++) {   /* every record is exactly its block_size */
                uint32_t bs;
                memcpy(&bs, p + offs[i], 4);
                if (offs[i + 1] - offs[i] != 4u + bs) { fprintf(stderr, "hostcheck: record %lld mis-indexed\n", (long long)(recs + i)); return 1; }
            }
            if (!quick) h = fnv(h, p, nbytes);
            recs += (uint64_t)n;
            bytes += nbytes;
            batches++;
        }
        const int32_t n_ref = rd ? bam_reader_header(rd)->n_ref : sam_reader_n_ref(sd);
        printf("alignments input=%s refs=%d records=%llu bytes=%llu digest=%016llx\n", is_bam ? "bam" : "sam", n_ref,
               (unsigned long long)recs, (unsigned long long)bytes, (unsigned long long)h);
        fprintf(stderr, "hostcheck: %llu batches\n", (unsigned long long)batches);
        if (rd) fprintf(stderr, "hostcheck: reader inflate stage %.3f s\n", bam_reader_inflate_seconds(rd));
        bam_reader_close(rd);
        sam_reader_close(sd);
    }
    return 0;
}
