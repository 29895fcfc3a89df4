#define _POSIX_C_SOURCE 200809L
/* TEST INFRASTRUCTURE -- see minibwa.h.  Plain C restatement of lh3/bwa's published
 * FM-index arithmetic (bwt.c / bntseq.c of bwa 0.7.17), written from the format. */
#include "minibwa.h"
#include <stdlib.h>
#include <string.h>

static __thread minibwa_counters_t g_cnt;
void minibwa_counters_get(minibwa_counters_t *out) { *out = g_cnt; }
void minibwa_counters_reset(void) { memset(&g_cnt, 0, sizeof g_cnt); }

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "minibwa: out of memory\n"); abort(); }
    return p;
}

static long file_size(FILE *fp) {
    long cur = ftell(fp), sz;
    fseek(fp, 0, SEEK_END);
    sz = ftell(fp);
    fseek(fp, cur, SEEK_SET);
    return sz;
}

static void read_all(void *dst, size_t bytes, FILE *fp, const char *fn) {
    if (bytes && fread(dst, 1, bytes, fp) != bytes) {
        fprintf(stderr, "minibwa: short read on %s\n", fn);
        abort();
    }
}

/* .bwt = u64 primary; u64 L2[1..4]; u32 bwt[bwt_size]  (counts already interleaved) */
bwt_t *bwt_restore_bwt(const char *fn) {
    FILE *fp = fopen(fn, "rb");
    if (!fp) return NULL;
    bwt_t *b = (bwt_t *)calloc(1, sizeof(bwt_t));
    long sz = file_size(fp);
    b->bwt_size = (bwtint_t)(sz - 5 * (long)sizeof(bwtint_t)) >> 2;
    b->bwt = (uint32_t *)xmalloc(b->bwt_size * 4);
    read_all(&b->primary, sizeof(bwtint_t), fp, fn);
    read_all(b->L2 + 1, 4 * sizeof(bwtint_t), fp, fn);
    read_all(b->bwt, b->bwt_size * 4, fp, fn);
    b->seq_len = b->L2[4];
    fclose(fp);
    return b;
}

/* .sa = u64 primary; u64 skipped[4]; u64 sa_intv; u64 seq_len; u64 sa[1..n_sa-1] */
void bwt_restore_sa(const char *fn, bwt_t *bwt) {
    if (!bwt) return;
    FILE *fp = fopen(fn, "rb");
    if (!fp) { fprintf(stderr, "minibwa: cannot open %s\n", fn); abort(); }
    bwtint_t primary, skipped[4], intv, seq_len;
    read_all(&primary, 8, fp, fn);
    read_all(skipped, 32, fp, fn);
    read_all(&intv, 8, fp, fn);
    read_all(&seq_len, 8, fp, fn);
    if (primary != bwt->primary || seq_len != bwt->seq_len) {
        fprintf(stderr, "minibwa: %s does not belong to this .bwt\n", fn);
        abort();
    }
    bwt->sa_intv = (int)intv;
    bwt->n_sa = (bwt->seq_len + intv) / intv;
    bwt->sa = (bwtint_t *)xmalloc(bwt->n_sa * sizeof(bwtint_t));
    bwt->sa[0] = (bwtint_t)-1;
    read_all(bwt->sa + 1, (bwt->n_sa - 1) * sizeof(bwtint_t), fp, fn);
    fclose(fp);
}

void bwt_destroy(bwt_t *bwt) {
    if (!bwt) return;
    free(bwt->sa);
    free(bwt->bwt);
    free(bwt);
}

/* count of base c among the 32 two-bit symbols packed in y */
static inline int occ32(uint64_t y, int c) {
    y = ((c & 2) ? y : ~y) >> 1 & ((c & 1) ? y : ~y) & 0x5555555555555555ull;
    return __builtin_popcountll(y);
}

/* number of c in BWT[0..k] (k is a row of the matrix WITH the sentinel row) */
bwtint_t bwt_occ(const bwt_t *bwt, bwtint_t k, ubyte_t c) {
    if (k == bwt->seq_len) return bwt->L2[c + 1] - bwt->L2[c];
    if (k == (bwtint_t)-1) return 0;
    k -= (k >= bwt->primary);  /* the sentinel is not stored */
    const uint32_t *p = bwt->bwt + ((k >> OCC_INTV_SHIFT) << 4);
    bwtint_t n = ((const bwtint_t *)p)[c];
    p += 8;  /* past the four u64 counters */
    bwtint_t full = (k >> 5) - ((k & ~(OCC_INTERVAL - 1)) >> 5); /* whole 32-base words */
    for (bwtint_t i = 0; i < full; ++i, p += 2) n += occ32((uint64_t)p[0] << 32 | p[1], c);
    uint64_t last = ((uint64_t)p[0] << 32 | p[1]) & ~((1ull << ((~k & 31) << 1)) - 1);
    n += occ32(last, c);
    if (c == 0) n -= ~k & 31;  /* the masked-off tail looked like A's */
    return n;
}

void bwt_2occ(const bwt_t *bwt, bwtint_t k, bwtint_t l, ubyte_t c, bwtint_t *ok, bwtint_t *ol) {
    g_cnt.n_2occ++;
    *ok = bwt_occ(bwt, k, c);
    *ol = bwt_occ(bwt, l, c);
}

static inline int bwt_B0(const bwt_t *bwt, bwtint_t k) {
    uint32_t w = bwt->bwt[((k >> OCC_INTV_SHIFT) << 4) + 8 + ((k & (OCC_INTERVAL - 1)) >> 4)];
    return w >> ((~k & 0xf) << 1) & 3;
}

static inline bwtint_t bwt_invPsi(const bwt_t *bwt, bwtint_t k) {
    if (k == bwt->primary) return 0;
    bwtint_t x = k - (k > bwt->primary);
    int c = bwt_B0(bwt, x);
    return bwt->L2[c] + bwt_occ(bwt, k, (ubyte_t)c);
}

bwtint_t bwt_sa(const bwt_t *bwt, bwtint_t k) {
    bwtint_t steps = 0, mask = (bwtint_t)bwt->sa_intv - 1;
    g_cnt.n_sa++;
    while (k & mask) {
        ++steps;
        g_cnt.n_lf++;
        k = bwt_invPsi(bwt, k);
    }
    return steps + bwt->sa[k / bwt->sa_intv];
}

/* .ann: "l_pac n_seqs seed" then per sequence "gi name anno" / "offset len n_ambs"
 * .amb: "l_pac n_seqs n_holes" then per hole "offset len char" */
bntseq_t *bns_restore(const char *prefix) {
    char fn[4096], line[8192];
    snprintf(fn, sizeof fn, "%s.ann", prefix);
    FILE *fp = fopen(fn, "r");
    if (!fp) return NULL;
    bntseq_t *bns = (bntseq_t *)calloc(1, sizeof(bntseq_t));
    long long l_pac; int n_seqs; unsigned seed;
    if (fscanf(fp, "%lld%d%u", &l_pac, &n_seqs, &seed) != 3) { fprintf(stderr, "minibwa: bad %s\n", fn); abort(); }
    bns->l_pac = l_pac; bns->n_seqs = n_seqs; bns->seed = seed;
    bns->anns = (bntann1_t *)calloc((size_t)n_seqs, sizeof(bntann1_t));
    for (int i = 0; i < n_seqs; ++i) {
        bntann1_t *a = bns->anns + i;
        char name[4096];
        if (fscanf(fp, "%u%4095s", &a->gi, name) != 2) { fprintf(stderr, "minibwa: bad %s\n", fn); abort(); }
        a->name = strdup(name);
        /* rest of the line (after one blank) is the free-text annotation */
        if (!fgets(line, sizeof line, fp)) line[0] = 0;
        char *s = line;
        while (*s == ' ' || *s == '\t') ++s;
        s[strcspn(s, "\r\n")] = 0;
        a->anno = strdup(strcmp(s, "(null)") == 0 ? "" : s);
        long long off; int len, n_ambs;
        if (fscanf(fp, "%lld%d%d", &off, &len, &n_ambs) != 3) { fprintf(stderr, "minibwa: bad %s\n", fn); abort(); }
        a->offset = off; a->len = len; a->n_ambs = n_ambs;
    }
    fclose(fp);

    snprintf(fn, sizeof fn, "%s.amb", prefix);
    fp = fopen(fn, "r");
    if (fp) {
        long long lp; int ns, nh;
        if (fscanf(fp, "%lld%d%d", &lp, &ns, &nh) == 3) {
            bns->n_holes = nh;
            bns->ambs = (bntamb1_t *)calloc((size_t)(nh ? nh : 1), sizeof(bntamb1_t));
            for (int i = 0; i < nh; ++i) {
                long long off; int len; char c[8];
                if (fscanf(fp, "%lld%d%7s", &off, &len, c) != 3) break;
                bns->ambs[i].offset = off; bns->ambs[i].len = len; bns->ambs[i].amb = c[0];
            }
        }
        fclose(fp);
    }
    snprintf(fn, sizeof fn, "%s.pac", prefix);
    bns->fp_pac = fopen(fn, "rb");  /* may be NULL; only load_pacseq reads it */
    return bns;
}

void bns_destroy(bntseq_t *bns) {
    if (!bns) return;
    if (bns->fp_pac) fclose(bns->fp_pac);
    for (int i = 0; i < bns->n_seqs; ++i) { free(bns->anns[i].name); free(bns->anns[i].anno); }
    free(bns->anns);
    free(bns->ambs);
    free(bns);
}

/* sequence id holding forward-strand coordinate pos_f, or -1 past the end */
int bns_pos2rid(const bntseq_t *bns, int64_t pos_f) {
    if (pos_f >= bns->l_pac) return -1;
    int left = 0, mid = 0, right = bns->n_seqs;
    while (left < right) {
        mid = (left + right) >> 1;
        if (pos_f >= bns->anns[mid].offset) {
            if (mid == bns->n_seqs - 1) break;
            if (pos_f < bns->anns[mid + 1].offset) break;
            left = mid + 1;
        } else {
            right = mid;
        }
    }
    return mid;
}

int bwa_idx_build(const char *fa, const char *prefix, int algo_type, int block_size) {
    (void)fa; (void)prefix; (void)algo_type; (void)block_size;
    fprintf(stderr, "minibwa: bwa_idx_build is not provided by the oracle (use tools/build_index.py)\n");
    return 1;
}

void err_fread_noeof(void *ptr, size_t size, size_t nmemb, FILE *stream) {
    if (fread(ptr, size, nmemb, stream) != nmemb) { fprintf(stderr, "minibwa: short fread\n"); abort(); }
}
