/*
 * pss-bam_amd/host/samline.c -- implementation of include/sam-parse.h.
 *
 * line2saml keeps the reference's contract (/root/reference/sam-parse.c:10-91) but is a
 * hand-written tokenizer: the eleven mandatory fields are split with the same rules one
 * sscanf("%s\t%u\t%s\t%lu\t%u\t%s\t%s\t%u\t%i\t%s\t%s") applies --
 *   - a field is a maximal run of non-white-space; ANY white space (not only TAB) separates
 *     fields and leading white space is skipped, because '\t' in a scanf format matches any
 *     amount of white space and %s/%u skip it anyway;
 *   - %u / %lu accept an optional sign and decimal digits (strtoul semantics, so "-1" wraps;
 *     the value is then truncated to the destination width like glibc does);
 *   - %i additionally understands 0x / 0 prefixes (strtol base 0);
 *   - a numeric field stops at its first non-digit: "12abc" yields 12 and the NEXT field then
 *     starts at "abc" (this is what shifts every later field in the reference, too);
 * -- without the per-call format interpretation, strlen passes and locale machinery of scanf.
 * Text fields longer than MAX_FIELD_WIDTH overflow the reference's buffers (undefined
 * behaviour there); here such a line is rejected (return 1).
 */
#include "sam-parse.h"

#include <errno.h>

static const char *skip_ws(const char *p)
{
    while (*p == ' ' || (*p >= '\t' && *p <= '\r')) p++;
    return p;
}

/* %s : returns NULL when no character is available (input failure) or the field does not fit */
static const char *take_str(const char *p, char *dst, size_t *len_out)
{
    size_t n = 0;
    p = skip_ws(p);
    while (*p && !(*p == ' ' || (*p >= '\t' && *p <= '\r'))) {
        if (n >= MAX_FIELD_WIDTH) return NULL;
        dst[n++] = *p++;
    }
    if (n == 0) return NULL;
    dst[n] = '\0';
    if (len_out) *len_out = n;
    return p;
}

/* %u / %lu */
static const char *take_ulong(const char *p, unsigned long *v)
{
    char *end;
    p = skip_ws(p);
    if (!*p) return NULL;
    errno = 0;
    *v = strtoul(p, &end, 10);
    if (end == p) return NULL; /* matching failure: no digits */
    return end;
}

/* %i */
static const char *take_int(const char *p, long *v)
{
    char *end;
    p = skip_ws(p);
    if (!*p) return NULL;
    errno = 0;
    *v = strtol(p, &end, 0);
    if (end == p) return NULL;
    return end;
}

int line2saml(const char *line, Saml *sp)
{
    const char *p = line;
    unsigned long u;
    long i;
    size_t seq_len = 0, qual_len = 0;

    if (!(p = take_str(p, sp->qname, NULL))) return 1;
    if (!(p = take_ulong(p, &u))) return 1;
    sp->flag = (unsigned int)u;
    if (!(p = take_str(p, sp->rname, NULL))) return 1;
    if (!(p = take_ulong(p, &u))) return 1;
    sp->pos = u;
    if (!(p = take_ulong(p, &u))) return 1;
    sp->mapq = (unsigned int)u;
    if (!(p = take_str(p, sp->cigar, NULL))) return 1;
    if (!(p = take_str(p, sp->mrnm, NULL))) return 1;
    if (!(p = take_ulong(p, &u))) return 1;
    sp->mpos = (unsigned int)u;
    if (!(p = take_int(p, &i))) return 1;
    sp->isize = (int)i;
    if (!(p = take_str(p, sp->seq, &seq_len))) return 1;
    if (!(p = take_str(p, sp->qual, &qual_len))) return 1;

    if (seq_len != qual_len) return 1; /* sam-parse.c:50 */
    sp->seq_len = (int)seq_len;

    sp->paired = (sp->flag >> 0) & 1;
    sp->proper_pair = (sp->flag >> 1) & 1;
    sp->unmap = (sp->flag >> 2) & 1;
    sp->munmap = (sp->flag >> 3) & 1;
    sp->reverse = (sp->flag >> 4) & 1;
    sp->mreverse = (sp->flag >> 5) & 1;
    sp->read1 = (sp->flag >> 6) & 1;
    sp->read2 = (sp->flag >> 7) & 1;
    sp->secondary = (sp->flag >> 8) & 1;
    sp->qc_failed = (sp->flag >> 9) & 1;
    sp->duplicate = (sp->flag >> 10) & 1;
    sp->supplementary = (sp->flag >> 11) & 1;

    if (!sp->paired) sp->isize = (int)seq_len; /* sam-parse.c:66-68 */

    /* optional fields: everything after the 11th TAB of the raw line (sam-parse.c:70-85);
     * left untouched when there is none, like the reference */
    {
        const char *q = line;
        int tabs = 0;
        while (*q && tabs < 11) {
            if (*q == '\t') tabs++;
            q++;
        }
        if (*q) {
            strncpy(sp->tags, q, MAX_FIELD_WIDTH);
            sp->tags[MAX_FIELD_WIDTH] = '\0';
        }
    }
    return 0;
}

int is_header(const char *line) { return line[0] == '@'; }

/* total length of the M operations; any other operation only advances the scan */
int aln_seq_len(const char *cigar)
{
    int total = 0;
    const char *p = cigar;
    while (*p) {
        char *end;
        long n = strtol(p, &end, 10);
        if (end == p || !*end) break; /* no number, or a number without an operation letter */
        if (*end == 'M') total += (int)n;
        p = end + 1;
    }
    return total;
}

int good_score(Saml *sp, float m, float b)
{
    if (sp->AS > 0) return (float)sp->AS >= (m * sp->seq_len) + b ? 1 : 0;
    return 1;
}
