/* xck_oracle.c - TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (libxck.so) never links or calls it.
 *
 * It restates, region by region and SNP by SNP exactly as the reference loops do, the
 * algorithm of hxj5/xcltk v0.5.2 on the same structure-of-arrays record batches the engine
 * consumes (include/xck.h), so that HIP results can be compared bit for bit:
 *
 *   basefc : xcltk/rdr/fc/core.py:69-178 (fc_features, fc_fet1, check_read,
 *            __get_include_frac/_len) and xcltk/rdr/fc/mcount.py:34-54,102-148
 *   BAF    : xcltk/baf/fc/core.py:42-247 (fc_features, fc_fet1, plp_snp, check_read),
 *            xcltk/baf/fc/mcount.py:39-60,109-150,206-256, xcltk/baf/fc/gfeature.py:33-39,
 *            xcltk/baf/fc/main.py:92-104 (SNP -> region join), xcltk/utils/sam.py:4-40
 *   fetch  : pysam/htslib semantics restated in SURVEY.md section 8c (third-party, absent from
 *            /root/reference): records with pos < stop and endpos > start, in BAM-list then
 *            file order.
 *
 * Pinning: the reference ships no tests.  This restatement is pinned by
 * (i) the two known-answer tests executed through the reference's own fc_fet1 (SURVEY 8c,
 * tests/test_oracle_kat.py) and (ii) golden .mtx files produced by running the unmodified
 * reference in the build container (oracle/refgen/, tests/golden/).
 *
 * Deliberately simple: per-region / per-SNP loops, qsort, tiny hash map.  No threads.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "../include/xck.h"

#define BAM_FPAIRED       1
#define BAM_FPROPER_PAIR  2
#define BAM_FUNMAP        4

typedef struct { int32_t b, i; } rref;                 /* (batch, index) of one read */

typedef struct {
    int64_t nnz[4];
    int32_t *row[4], *col[4], *val[4];
    int64_t cap[4];
    int64_t n_reads, n_pass;
} xo_result;

typedef struct { int32_t cell; uint64_t umi; } cu_t;
typedef struct { int32_t cell; uint64_t umi; uint8_t bits; } cub_t;
typedef struct { int32_t cell; uint64_t umi; int8_t allele; } cua_t;   /* allele: nibble 0..15 or -1 */

/* ---------------------------------------------------------------- helpers */
static void coo_push(xo_result *r, int m, int32_t row, int32_t col, int32_t val) {
    if (r->nnz[m] == r->cap[m]) {
        r->cap[m] = r->cap[m] ? r->cap[m] * 2 : 1024;
        r->row[m] = (int32_t*)realloc(r->row[m], sizeof(int32_t) * r->cap[m]);
        r->col[m] = (int32_t*)realloc(r->col[m], sizeof(int32_t) * r->cap[m]);
        r->val[m] = (int32_t*)realloc(r->val[m], sizeof(int32_t) * r->cap[m]);
    }
    r->row[m][r->nnz[m]] = row; r->col[m][r->nnz[m]] = col; r->val[m][r->nnz[m]] = val;
    r->nnz[m]++;
}

static int consumes_ref(int op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
static int is_aligned(int op)   { return op == 0 || op == 7 || op == 8; }

/* htslib bam_endpos() */
static int32_t read_endpos(const xck_batch *b, int i) {
    int32_t rlen = 0;
    uint32_t c0 = b->cig_off[i], c1 = b->cig_off[i + 1];
    if (!(b->flag[i] & BAM_FUNMAP) && c1 > c0) {
        for (uint32_t c = c0; c < c1; c++)
            if (consumes_ref(b->cigar[c] & 0xF)) rlen += (int32_t)(b->cigar[c] >> 4);
    } else rlen = 1;
    if (rlen == 0) rlen = 1;
    return b->pos[i] + rlen;
}

/* len(read.positions): number of M/=/X bases */
static int32_t n_aligned(const xck_batch *b, int i) {
    int32_t n = 0;
    for (uint32_t c = b->cig_off[i]; c < b->cig_off[i + 1]; c++)
        if (is_aligned(b->cigar[c] & 0xF)) n += (int32_t)(b->cigar[c] >> 4);
    return n;
}

/* __get_include_len(): aligned positions p with s <= p <= e (0-based inclusive) */
static int32_t n_included(const xck_batch *b, int i, int64_t s, int64_t e) {
    int64_t p = b->pos[i];
    int32_t m = 0;
    for (uint32_t c = b->cig_off[i]; c < b->cig_off[i + 1]; c++) {
        int op = b->cigar[c] & 0xF;
        int64_t l = b->cigar[c] >> 4;
        if (is_aligned(op)) {
            for (int64_t k = 0; k < l; k++) if (s <= p + k && p + k <= e) m++;
            p += l;
        } else if (consumes_ref(op)) p += l;
    }
    return m;
}

/* check_read(), rdr/fc/core.py:46-62 == baf/fc/core.py:18-34.  Cell / UMI tag presence is
 * folded into cell < 0 / umi == NONE by the decoder; both only ever drop the read. */
static int check_read(const xck_config *cf, const xck_batch *b, int i) {
    uint32_t flag = b->flag[i];
    if ((double)b->mapq[i] < cf->min_mapq) return -2;
    if (cf->excl_flag && (flag & cf->excl_flag)) return -3;
    if (cf->incl_flag && !(flag & cf->incl_flag)) return -4;
    if (cf->no_orphan && (flag & BAM_FPAIRED) && !(flag & BAM_FPROPER_PAIR)) return -5;
    if (b->cell[i] < 0) return -11;           /* tag missing, or cell not in list (mcount :124-127) */
    if (b->umi[i] == XCK_UMI_NONE) return -12; /* tag missing, or empty key (mcount :41)            */
    if (n_aligned(b, i) < cf->min_len) return -21;
    return 0;
}

/* UCount.push_read(), baf/fc/mcount.py:39-60 + get_query_bases(), utils/sam.py:4-40:
 * nibble of the query base aligned to 0-based reference position p0, or -1. */
static int allele_at(const xck_batch *b, int i, int64_t p0) {
    int64_t r = b->pos[i], q = 0;
    for (uint32_t c = b->cig_off[i]; c < b->cig_off[i + 1]; c++) {
        int op = b->cigar[c] & 0xF;
        int64_t l = b->cigar[c] >> 4;
        if (is_aligned(op)) {
            if (p0 >= r && p0 < r + l) {
                int64_t qi = q + (p0 - r);
                uint32_t s0 = b->seq_off[i], s1 = b->seq_off[i + 1];
                if ((uint64_t)(qi >> 1) >= (uint64_t)(s1 - s0)) return -1;   /* reference would raise */
                uint8_t by = b->seq[s0 + (qi >> 1)];
                return (qi & 1) ? (by & 0xF) : (by >> 4);
            }
            r += l; q += l;
        } else if (op == 1 || op == 4) q += l;       /* I, S advance the query only */
        else if (consumes_ref(op)) r += l;           /* D, N advance the reference only */
    }
    return -1;
}

static int cmp_cu(const void *a, const void *b) {
    const cu_t *x = (const cu_t*)a, *y = (const cu_t*)b;
    if (x->cell != y->cell) return x->cell < y->cell ? -1 : 1;
    if (x->umi != y->umi) return x->umi < y->umi ? -1 : 1;
    return 0;
}
static int cmp_cub(const void *a, const void *b) {
    const cub_t *x = (const cub_t*)a, *y = (const cub_t*)b;
    if (x->cell != y->cell) return x->cell < y->cell ? -1 : 1;
    if (x->umi != y->umi) return x->umi < y->umi ? -1 : 1;
    return 0;
}

/* ---------------------------------------------------------------- per-contig fetch index */
typedef struct {
    rref   *reads;      /* all reads of the contig, sorted by ordinal (BAM-list, then file order) */
    int64_t n;
    int64_t *seg_beg;   /* start of each BAM's run inside reads[] */
    int32_t n_seg;
    int32_t maxspan;    /* max(endpos - pos) */
    int     sorted;     /* every BAM run is sorted by pos */
} contig_idx;

static const xck_batch *g_batches;
static int cmp_ord(const void *a, const void *b) {
    const rref *x = (const rref*)a, *y = (const rref*)b;
    uint64_t ox = g_batches[x->b].ordinal_base + (uint64_t)x->i;
    uint64_t oy = g_batches[y->b].ordinal_base + (uint64_t)y->i;
    return ox < oy ? -1 : (ox > oy ? 1 : 0);
}

static void build_index(const xck_config *cf, const xck_batch *bt, int nb, contig_idx *ci) {
    for (int c = 0; c < cf->n_contigs; c++) { memset(&ci[c], 0, sizeof(ci[c])); ci[c].sorted = 1; }
    for (int b = 0; b < nb; b++) if (bt[b].contig >= 0 && bt[b].contig < cf->n_contigs) ci[bt[b].contig].n += bt[b].n_reads;
    for (int c = 0; c < cf->n_contigs; c++) { ci[c].reads = (rref*)malloc(sizeof(rref) * (ci[c].n + 1)); ci[c].n = 0; }
    for (int b = 0; b < nb; b++) {
        int c = bt[b].contig;
        if (c < 0 || c >= cf->n_contigs) continue;
        for (int i = 0; i < bt[b].n_reads; i++) { ci[c].reads[ci[c].n].b = b; ci[c].reads[ci[c].n].i = i; ci[c].n++; }
    }
    g_batches = bt;
    for (int c = 0; c < cf->n_contigs; c++) {
        contig_idx *x = &ci[c];
        qsort(x->reads, x->n, sizeof(rref), cmp_ord);
        x->seg_beg = (int64_t*)malloc(sizeof(int64_t) * (x->n + 2));
        x->n_seg = 0;
        uint64_t cur = ~0ull;
        for (int64_t k = 0; k < x->n; k++) {
            const xck_batch *b = &bt[x->reads[k].b];
            int i = x->reads[k].i;
            uint64_t bam = (b->ordinal_base + (uint64_t)i) >> 40;
            if (bam != cur) { x->seg_beg[x->n_seg++] = k; cur = bam; }
            else {
                const xck_batch *pb = &bt[x->reads[k - 1].b];
                if (pb->pos[x->reads[k - 1].i] > b->pos[i]) x->sorted = 0;
            }
            int32_t span = read_endpos(b, i) - b->pos[i];
            if (span > x->maxspan) x->maxspan = span;
        }
        x->seg_beg[x->n_seg] = x->n;
    }
}

/* iterate reads of pysam fetch(contig, start0, stop0) : pos < stop0 && endpos > start0 */
#define FETCH_BEGIN(ci_, bt_, start0_, stop0_)                                              \
    for (int seg_ = 0; seg_ < (ci_)->n_seg; seg_++) {                                        \
        int64_t lo_ = (ci_)->seg_beg[seg_], hi_ = (ci_)->seg_beg[seg_ + 1];                  \
        if ((ci_)->sorted) {                                                                 \
            int64_t a_ = lo_, z_ = hi_, want_ = (int64_t)(start0_) - (ci_)->maxspan;          \
            while (a_ < z_) { int64_t m_ = (a_ + z_) >> 1;                                   \
                if ((bt_)[(ci_)->reads[m_].b].pos[(ci_)->reads[m_].i] < want_) a_ = m_ + 1; else z_ = m_; } \
            lo_ = a_;                                                                        \
        }                                                                                    \
        for (int64_t k_ = lo_; k_ < hi_; k_++) {                                             \
            const xck_batch *B = &(bt_)[(ci_)->reads[k_].b]; int I = (ci_)->reads[k_].i;     \
            if ((ci_)->sorted && B->pos[I] >= (stop0_)) break;                               \
            if (!(B->pos[I] < (stop0_) && read_endpos(B, I) > (start0_))) continue;
#define FETCH_END }}

/* ---------------------------------------------------------------- basefc */
static void run_basefc(const xck_config *cf, const xck_batch *bt, int nb, contig_idx *ci, xo_result *out, int g_lo, int g_hi) {
    int frac_mode = (cf->min_include > 0.0 && cf->min_include < 1.0);
    cu_t *set = NULL; int64_t cap = 0;
    (void)nb;
    for (int g = g_lo; g < g_hi; g++) {
        const xck_region *rg = &cf->regions[g];
        if (rg->contig < 0 || rg->contig >= cf->n_contigs) continue;
        /* reg.start = start, reg.end = end_incl + 1 (rdr/fc/utils.py:42).
         * sam_fetch(chrom, reg.start, reg.end - 1) -> fetch(start - 1, end_incl)  (utils/sam.py:105) */
        int64_t start0 = (int64_t)rg->start - 1, stop0 = rg->end;
        if (start0 < 0 || start0 > stop0) continue;           /* pysam raises -> region gets 0 */
        int64_t s = (int64_t)rg->start - 1, e = (int64_t)rg->end - 1;   /* reg.start-1, reg.end-2 */
        int64_t n = 0;
        contig_idx *x = &ci[rg->contig];
        FETCH_BEGIN(x, bt, start0, stop0)
            if (check_read(cf, B, I) < 0) continue;
            int32_t m = n_included(B, I, s, e);
            if (frac_mode) {
                int32_t na = n_aligned(B, I);
                if (na <= 0) continue;                        /* reference: None < float raises */
                if ((double)m / (double)na < cf->min_include) continue;
            } else {
                if ((double)m < cf->min_include) continue;
            }
            if (n == cap) { cap = cap ? cap * 2 : 4096; set = (cu_t*)realloc(set, sizeof(cu_t) * cap); }
            set[n].cell = B->cell[I]; set[n].umi = B->umi[I]; n++;
        FETCH_END
        if (!n) continue;
        qsort(set, n, sizeof(cu_t), cmp_cu);
        int64_t k = 0;
        while (k < n) {                                       /* count = len(umi_set) per cell */
            int32_t cell = set[k].cell, cnt = 0;
            while (k < n && set[k].cell == cell) {
                uint64_t u = set[k].umi; cnt++;
                while (k < n && set[k].cell == cell && set[k].umi == u) k++;
            }
            coo_push(out, 0, g, cell, cnt);
        }
    }
    free(set);
}

/* ---------------------------------------------------------------- BAF */
typedef struct { cua_t *e; int64_t n; int filtered; } snp_plp;

static uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

/* plp_snp(), baf/fc/core.py:198-247 */
static void pileup_snp(const xck_config *cf, const xck_batch *bt, contig_idx *ci, const xck_snp *sn, snp_plp *out) {
    static const char NT16[] = "=ACMGRSVTWYHKDBN";
    out->e = NULL; out->n = 0; out->filtered = 0;
    if (sn->contig < 0 || sn->contig >= cf->n_contigs || sn->pos < 1) { out->filtered = 1; return; }
    contig_idx *x = &ci[sn->contig];
    int64_t cap = 64, n = 0;
    cua_t *lst = (cua_t*)malloc(sizeof(cua_t) * cap);
    int64_t hcap = 256; int64_t *ht = (int64_t*)malloc(sizeof(int64_t) * hcap);
    for (int64_t k = 0; k < hcap; k++) ht[k] = -1;
    int64_t start0 = (int64_t)sn->pos - 1, stop0 = sn->pos;   /* sam_fetch(chrom, pos, pos) */
    FETCH_BEGIN(x, bt, start0, stop0)
        if (check_read(cf, B, I) < 0) continue;
        int32_t cell = B->cell[I]; uint64_t umi = B->umi[I];
        /* SCount.push_read(): only the first read of a (cell, UMI) is used (mcount.py:118-119) */
        uint64_t h = mix64(umi * 0x9E3779B97F4A7C15ull + (uint64_t)(uint32_t)cell) & (uint64_t)(hcap - 1);
        int found = 0;
        while (ht[h] >= 0) { if (lst[ht[h]].cell == cell && lst[ht[h]].umi == umi) { found = 1; break; } h = (h + 1) & (uint64_t)(hcap - 1); }
        if (found) continue;
        if (n == cap) { cap *= 2; lst = (cua_t*)realloc(lst, sizeof(cua_t) * cap); }
        lst[n].cell = cell; lst[n].umi = umi; lst[n].allele = (int8_t)allele_at(B, I, start0);
        ht[h] = n; n++;
        if (n * 2 > hcap) {                                   /* grow + rehash */
            hcap *= 2; ht = (int64_t*)realloc(ht, sizeof(int64_t) * hcap);
            for (int64_t k = 0; k < hcap; k++) ht[k] = -1;
            for (int64_t k = 0; k < n; k++) {
                uint64_t hh = mix64(lst[k].umi * 0x9E3779B97F4A7C15ull + (uint64_t)(uint32_t)lst[k].cell) & (uint64_t)(hcap - 1);
                while (ht[hh] >= 0) hh = (hh + 1) & (uint64_t)(hcap - 1);
                ht[hh] = k;
            }
        }
    FETCH_END
    free(ht);
    /* MCount.stat(): tallies A,C,G,T,N over UMIs with an allele (mcount.py:140-150,250-256) */
    int64_t tc[5] = {0, 0, 0, 0, 0};
    for (int64_t k = 0; k < n; k++) {
        if (lst[k].allele < 0) continue;
        char ch = NT16[lst[k].allele];
        int idx = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
        tc[idx]++;
    }
    int64_t tot = tc[0] + tc[1] + tc[2] + tc[3] + tc[4];
    int ri = sn->ref == 'A' ? 0 : sn->ref == 'C' ? 1 : sn->ref == 'G' ? 2 : sn->ref == 'T' ? 3 : 4;
    int ai = sn->alt == 'A' ? 0 : sn->alt == 'C' ? 1 : sn->alt == 'G' ? 2 : sn->alt == 'T' ? 3 : 4;
    if ((double)tot < cf->min_count) out->filtered = 1;                    /* core.py:239-241 */
    else {
        int64_t minor = tc[ri] < tc[ai] ? tc[ri] : tc[ai];
        if ((double)minor < (double)tot * cf->min_maf) out->filtered = 1;  /* core.py:242-246 */
    }
    out->e = lst; out->n = n;
}

static const xck_snp *g_snps;
static int cmp_snp_idx(const void *a, const void *b) {
    const xck_snp *x = &g_snps[*(const int*)a], *y = &g_snps[*(const int*)b];
    if (x->contig != y->contig) return x->contig < y->contig ? -1 : 1;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return *(const int*)a < *(const int*)b ? -1 : 1;
}

/* fc_fet1() over the regions [g_lo, g_hi) given every SNP's pileup (plp) and the SNPs ordered by (contig, pos) (sidx) */
static void baf_regions(const xck_config *cf, const snp_plp *plp, const int *sidx, xo_result *out, int g_lo, int g_hi) {
    static const char NT16[] = "=ACMGRSVTWYHKDBN";
    cub_t *set = NULL; int64_t cap = 0;
    for (int g = g_lo; g < g_hi; g++) {
        const xck_region *rg = &cf->regions[g];
        int64_t n = 0;
        /* snp_set.fetch(reg.chrom, reg.start, reg.end): start <= pos <= end_incl (main.py:93);
         * sidx[] lists SNPs ordered by (contig, pos) so the scan can start at a lower bound. */
        int64_t a = 0, z = cf->n_snps;
        while (a < z) { int64_t mid = (a + z) >> 1; const xck_snp *q = &cf->snps[sidx[mid]];
            if (q->contig < rg->contig || (q->contig == rg->contig && q->pos < rg->start)) a = mid + 1; else z = mid; }
        for (int64_t si = a; si < cf->n_snps; si++) {
            int s = sidx[si];
            const xck_snp *sn = &cf->snps[s];
            if (sn->contig != rg->contig || sn->pos > rg->end) break;
            if (plp[s].filtered) continue;
            /* SNPs that region-wise local phasing dropped from THIS region's list (baf/fc/phasing.py:44-49) */
            { int skip = 0; for (int x = 0; x < cf->n_excl_pairs; x++) if (cf->excl_region[x] == g && cf->excl_snp[x] == s) { skip = 1; break; } if (skip) continue; }
            for (int64_t k = 0; k < plp[s].n; k++) {
                const cua_t *u = &plp[s].e[k];
                if (u->allele < 0) continue;
                char ch = NT16[u->allele];
                /* snp.gt = {ref: ref_idx, alt: alt_idx} (gfeature.py:33): alt wins if ref == alt */
                int idx = -1;
                if (ch == (char)sn->ref) idx = sn->ref_hap;
                if (ch == (char)sn->alt) idx = sn->alt_hap;
                if (n == cap) { cap = cap ? cap * 2 : 4096; set = (cub_t*)realloc(set, sizeof(cub_t) * cap); }
                set[n].cell = u->cell; set[n].umi = u->umi;
                set[n].bits = idx == 0 ? 1 : idx == 1 ? 2 : 4;
                n++;
            }
        }
        if (!n) continue;
        qsort(set, n, sizeof(cub_t), cmp_cub);
        int64_t k = 0;
        while (k < n) {
            int32_t cell = set[k].cell;
            int32_t ref = 0, alt = 0, uni = 0, oth = 0;
            while (k < n && set[k].cell == cell) {
                uint64_t u = set[k].umi; int bits = 0;
                while (k < n && set[k].cell == cell && set[k].umi == u) { bits |= set[k].bits; k++; }
                if (bits & 1) ref++;
                if (bits & 2) alt++;
                if (bits & 3) uni++;
                else if (bits & 4) oth++;
            }
            int32_t dp = uni;
            if (ref + alt != dp) {                            /* core.py:181-192 */
                if (cf->no_dup_hap) { int32_t share = ref + alt - dp; ref -= share; alt -= share; }
                dp = ref + alt;
            }
            if (dp + oth <= 0) continue;                      /* core.py:89-90 */
            if (alt > 0) coo_push(out, 1, g, cell, alt);
            if (dp > 0)  coo_push(out, 2, g, cell, dp);
            if (oth > 0) coo_push(out, 3, g, cell, oth);
        }
    }
    free(set);
}

static int *sorted_snp_index(const xck_config *cf) {
    int *sidx = (int*)malloc(sizeof(int) * (cf->n_snps + 1));
    for (int s = 0; s < cf->n_snps; s++) sidx[s] = s;
    g_snps = cf->snps;
    qsort(sidx, cf->n_snps, sizeof(int), cmp_snp_idx);
    return sidx;
}

static void run_baf(const xck_config *cf, const xck_batch *bt, int nb, contig_idx *ci, xo_result *out) {
    (void)nb;
    snp_plp *plp = (snp_plp*)calloc(cf->n_snps ? cf->n_snps : 1, sizeof(snp_plp));
    for (int s = 0; s < cf->n_snps; s++) pileup_snp(cf, bt, ci, &cf->snps[s], &plp[s]);
    int *sidx = sorted_snp_index(cf);
    baf_regions(cf, plp, sidx, out, 0, cf->n_regions);
    free(sidx);
    for (int s = 0; s < cf->n_snps; s++) free(plp[s].e);
    free(plp);
}

/* ---------------------------------------------------------------- entry points */
void xo_free(xo_result *r);
int xo_run(const xck_config *cf, const xck_batch *batches, int n_batches, xo_result *out) {
    memset(out, 0, sizeof(*out));
    contig_idx *ci = (contig_idx*)calloc(cf->n_contigs ? cf->n_contigs : 1, sizeof(contig_idx));
    build_index(cf, batches, n_batches, ci);
    for (int b = 0; b < n_batches; b++) out->n_reads += batches[b].n_reads;
    if (cf->mode == XCK_MODE_BASEFC) run_basefc(cf, batches, n_batches, ci, out, 0, cf->n_regions);
    else if (cf->mode == XCK_MODE_BAF) run_baf(cf, batches, n_batches, ci, out);
    else { free(ci); return -1; }
    for (int c = 0; c < cf->n_contigs; c++) { free(ci[c].reads); free(ci[c].seg_beg); }
    free(ci);
    return 0;
}

/* The same work on n_threads host threads, split the way the reference splits it (fc_core / afc_core, rdr/fc/main.py:196-208):
 * contiguous chunks of regions per worker, results concatenated in chunk order - here many small chunks handed out
 * dynamically, because a few genes hold most reads.  The per-SNP pileups of the BAF path (independent of each other) run in
 * parallel first.  Output is identical to xo_run() (tests/test_oracle_kat.py).  Used for bench.py's cpu_baseline. */
int xo_run_mt(const xck_config *cf, const xck_batch *batches, int n_batches, xo_result *out, int n_threads) {
    if (n_threads <= 1) return xo_run(cf, batches, n_batches, out);
    if (cf->mode != XCK_MODE_BASEFC && cf->mode != XCK_MODE_BAF) return -1;
    memset(out, 0, sizeof(*out));
    contig_idx *ci = (contig_idx*)calloc(cf->n_contigs ? cf->n_contigs : 1, sizeof(contig_idx));
    build_index(cf, batches, n_batches, ci);
    for (int b = 0; b < n_batches; b++) out->n_reads += batches[b].n_reads;
    const int G = cf->n_regions;
    int n_chunk = n_threads * 16; if (n_chunk > G) n_chunk = G > 0 ? G : 1;
    xo_result *part = (xo_result*)calloc(n_chunk, sizeof(xo_result));
    snp_plp *plp = NULL; int *sidx = NULL;
    if (cf->mode == XCK_MODE_BAF) {
        plp = (snp_plp*)calloc(cf->n_snps ? cf->n_snps : 1, sizeof(snp_plp));
        #pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads)
        for (int s = 0; s < cf->n_snps; s++) pileup_snp(cf, batches, ci, &cf->snps[s], &plp[s]);
        sidx = sorted_snp_index(cf);
    }
    #pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int c = 0; c < n_chunk; c++) {
        const int g_lo = (int)((int64_t)G * c / n_chunk), g_hi = (int)((int64_t)G * (c + 1) / n_chunk);
        if (cf->mode == XCK_MODE_BASEFC) run_basefc(cf, batches, n_batches, ci, &part[c], g_lo, g_hi);
        else baf_regions(cf, plp, sidx, &part[c], g_lo, g_hi);
    }
    for (int m = 0; m < 4; m++) {
        int64_t tot = 0;
        for (int c = 0; c < n_chunk; c++) tot += part[c].nnz[m];
        out->nnz[m] = out->cap[m] = tot;
        if (!tot) continue;
        out->row[m] = (int32_t*)malloc(sizeof(int32_t) * tot); out->col[m] = (int32_t*)malloc(sizeof(int32_t) * tot); out->val[m] = (int32_t*)malloc(sizeof(int32_t) * tot);
        int64_t at = 0;
        for (int c = 0; c < n_chunk; c++) {
            const int64_t k = part[c].nnz[m];
            if (k) { memcpy(out->row[m] + at, part[c].row[m], sizeof(int32_t) * k); memcpy(out->col[m] + at, part[c].col[m], sizeof(int32_t) * k);
                     memcpy(out->val[m] + at, part[c].val[m], sizeof(int32_t) * k); at += k; }
        }
    }
    for (int c = 0; c < n_chunk; c++) xo_free(&part[c]);
    free(part);
    if (plp) { for (int s = 0; s < cf->n_snps; s++) free(plp[s].e); free(plp); }
    free(sidx);
    for (int c = 0; c < cf->n_contigs; c++) { free(ci[c].reads); free(ci[c].seg_beg); }
    free(ci);
    return 0;
}

void xo_free(xo_result *r) {
    for (int m = 0; m < 4; m++) { free(r->row[m]); free(r->col[m]); free(r->val[m]); }
    memset(r, 0, sizeof(*r));
}

/* m/float(n) < v as the reference evaluates it; exported so tests can pin the GPU's
 * double-precision divide against the host's (rdr/fc/core.py:37). */
int xo_frac_drop(int32_t m, int32_t n, double v) { return ((double)m / (double)n) < v; }
