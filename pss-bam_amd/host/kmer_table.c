/*
 * pss-bam_amd/host/kmer_table.c -- implementation of include/kmer.h.
 *
 * Same data structure contract as the reference (/root/reference/kmer.c): a 4^K_AR_SIZE
 * array of node pointers indexed by the first min(k, K_AR_SIZE) bases, 4-ary nodes below
 * it for longer k-mers, saturating unsigned counts.  On the GPU the counter is a flat 4^k
 * histogram; the fragkon front end reads it directly, so this module mainly serves callers
 * of the reference API -- and `ksp_add_count` lets a flat histogram be poured into a KSP.
 */
#include "kmer.h"

static inline int base2bits(char c)
{
    switch (toupper((unsigned char)c)) { /* case-folded, kmer.c:189 */
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
    }
}

KSP init_KSP(int k)
{
    size_t n = (size_t)1 << (K_AR_SIZE * 2);
    KSP ks = (KSP)malloc(sizeof(Kmers));
    if (!ks) return NULL;
    ks->k = (size_t)k;
    ks->k_ar_size = K_AR_SIZE;
    ks->ka = (ktnP *)calloc(n, sizeof(ktnP));
    if (!ks->ka) { free(ks); return NULL; }
    return ks;
}

int kmer2inx(const char *kmer, const size_t kmer_len, size_t *inx)
{
    size_t v = 0;
    for (size_t i = 0; i < kmer_len; i++) {
        int b = base2bits(kmer[i]);
        if (b < 0) return 0;
        v = (v << 2) | (size_t)b;
    }
    *inx = v;
    return 1;
}

static ktnP new_node(void) { return (ktnP)calloc(1, sizeof(ktn)); }

static ktnP *child_slot(ktnP n, int b)
{
    switch (b) {
    case 0: return &n->Ap;
    case 1: return &n->Cp;
    case 2: return &n->Gp;
    default: return &n->Tp;
    }
}

/* node holding the count of `kmer`, created on demand when `create`; NULL = invalid / absent */
static ktnP locate(const char *kmer, KSP ks, int create)
{
    size_t head = ks->k < ks->k_ar_size ? ks->k : ks->k_ar_size, inx;
    ktnP cur;
    if (!kmer2inx(kmer, head, &inx)) return NULL;
    /* validate the tail before touching the tree so an invalid k-mer allocates nothing */
    for (size_t i = head; i < ks->k; i++)
        if (base2bits(kmer[i]) < 0) return NULL;
    cur = ks->ka[inx];
    if (!cur) {
        if (!create) return NULL;
        cur = ks->ka[inx] = new_node();
        if (!cur) return NULL;
    }
    for (size_t i = head; i < ks->k; i++) {
        ktnP *slot = child_slot(cur, base2bits(kmer[i]));
        if (!*slot) {
            if (!create) return NULL;
            *slot = new_node();
            if (!*slot) return NULL;
        }
        cur = *slot;
    }
    return cur;
}

int add_to_ksp(const char *kmer, KSP ks)
{
    ktnP n = locate(kmer, ks, 1);
    if (!n) return -1;
    if (n->count < UINT_MAX) n->count += 1; /* sticks at UINT_MAX, kmer.c:102-104 */
    return 0;
}

unsigned int kmer2count(const char *kmer, const KSP ks)
{
    ktnP n = locate(kmer, ks, 0);
    return n ? n->count : 0;
}

static void free_tree(ktnP n)
{
    if (!n) return;
    free_tree(n->Ap);
    free_tree(n->Cp);
    free_tree(n->Gp);
    free_tree(n->Tp);
    free(n);
}

int destroy_KSP(KSP ks)
{
    if (!ks) return 0;
    for (size_t i = 0; i < ((size_t)1 << (K_AR_SIZE * 2)); i++) free_tree(ks->ka[i]);
    free(ks->ka);
    free(ks);
    return 0;
}
