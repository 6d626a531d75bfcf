/*
 * pss-bam_amd/host/bam2sam_main.c -- `bam2sam [-r RG] file.bam`: prints the alignment records
 * of a BAM as SAM text, like `samtools view` (no header).  It exists so the test suite can
 * feed the UNMODIFIED reference binaries (which popen "samtools view") the same BAM the
 * engine consumes; see oracle/shim/samtools.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bam_reader.h"

int main(int argc, char **argv)
{
    const char *rg = NULL, *path = NULL;
    char err[512];
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-r") && i + 1 < argc) rg = argv[++i];
        else path = argv[i];
    }
    if (!path) {
        fprintf(stderr, "usage: bam2sam [-r READGROUP] file.bam\n");
        return 2;
    }
    bam_reader *rd = bam_reader_open(path, 0, 0, err, sizeof err);
    if (!rd) {
        fprintf(stderr, "bam2sam: %s\n", err);
        return 1;
    }
    const bam_header *h = bam_reader_header(rd);
    size_t cap = 1u << 20;
    char *line = (char *)malloc(cap);
    static char iobuf[1 << 20];
    setvbuf(stdout, iobuf, _IOFBF, sizeof iobuf);
    for (;;) {
        const uint8_t *recs;
        const uint32_t *offs;
        size_t nbytes;
        int64_t n = bam_reader_next(rd, &recs, &offs, &nbytes);
        if (n < 0) {
            fprintf(stderr, "bam2sam: %s\n", bam_reader_error(rd));
            return 1;
        }
        if (n == 0) break;
        for (int64_t i = 0; i < n; i++) {
            const uint8_t *rec = recs + offs[i];
            uint32_t len = offs[i + 1] - offs[i];
            if (rg && !bam_record_has_rg(rec, len, rg)) continue;
            long w;
            while ((w = bam_record_to_sam(rec, len, h, line, cap)) < 0 && cap < ((size_t)1 << 30)) {
                cap *= 2;
                line = (char *)realloc(line, cap);
            }
            if (w < 0) {
                fprintf(stderr, "bam2sam: malformed record\n");
                return 1;
            }
            fwrite(line, 1, (size_t)w, stdout);
        }
    }
    free(line);
    bam_reader_close(rd);
    return 0;
}
