// api.cpp - C-ABI entry points of libxck.so (include/xck.h) except the BAM functions (bam.cpp).
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstring>
#include <thread>
#include "xck_internal.h"

using namespace xck;

extern "C" {

const char* xck_version(void) { return XCK_VERSION_STR; }
int xck_abi_version(void) { return XCK_ABI_VERSION; }

int xck_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* xck_last_error(const xck_engine* e) { return e ? e->err.c_str() : get_thread_error(); }

int xck_create(const xck_config* cfg, xck_engine** out) {
    if (!cfg || !out) { set_thread_error("null argument"); return XCK_E_ARG; }
    *out = nullptr;
    if (cfg->struct_size != sizeof(xck_config)) { set_thread_error("xck_config.struct_size mismatch (ABI)"); return XCK_E_ARG; }
    if (cfg->mode != XCK_MODE_BASEFC && cfg->mode != XCK_MODE_BAF && cfg->mode != XCK_MODE_BOTH) { set_thread_error("invalid mode"); return XCK_E_ARG; }
    if (cfg->n_cells <= 0 || cfg->n_contigs < 0 || cfg->n_regions < 0 || cfg->n_snps < 0) { set_thread_error("invalid table sizes"); return XCK_E_ARG; }
    if ((cfg->n_regions > 0 && !cfg->regions) || (cfg->n_snps > 0 && !cfg->snps)) { set_thread_error("null table pointer"); return XCK_E_ARG; }
    xck_engine* e = new xck_engine();
    e->umi_bits = key_layout(cfg).ubits;
    e->mode = cfg->mode;
    if (!(cfg->flags & XCK_F_DECODE_ONLY)) {
        const int modes[2] = { cfg->mode == XCK_MODE_BOTH ? XCK_MODE_BASEFC : cfg->mode, XCK_MODE_BAF };
        const int n = cfg->mode == XCK_MODE_BOTH ? 2 : 1;
        for (int k = 0; k < n; k++) {
            xck_config c = *cfg;
            c.mode = modes[k];
            if (n == 2) c.flags |= XCK_F_LAYOUT_BOTH;          // one key layout -> one decode serves both pipelines
            e->impl = nullptr;
            int rc = engine_create(&c, e);
            e->impls[k] = e->impl; e->n_impl = k + 1;
            if (rc) { set_thread_error(e->err); xck_destroy(e); return rc; }
        }
        e->impl = e->impls[0];
    }
    // decoder settings
    DecodeCfg& d = e->dec;
    d.use_barcodes = cfg->barcodes != nullptr;
    if (d.use_barcodes) { d.build_barcodes(cfg->barcodes, cfg->n_cells); d.cell_tag[0] = cfg->cell_tag[0]; d.cell_tag[1] = cfg->cell_tag[1]; }
    d.use_umi = cfg->umi_tag[0] != 0;
    if (d.use_umi) { d.umi_tag[0] = cfg->umi_tag[0]; d.umi_tag[1] = cfg->umi_tag[1]; }
    d.umi_bits = e->umi_bits;
    d.want_seq = (cfg->mode & XCK_MODE_BAF) != 0;
    d.verify_crc = (cfg->flags & XCK_F_VERIFY_CRC) != 0;
    d.n_threads = cfg->n_threads;
    d.max_batch_reads = cfg->max_batch_reads;
    *out = e;
    return XCK_OK;
}

#define FOR_IMPLS(e, expr) do { if ((e)->n_impl == 0) { (e)->impl = nullptr; int rc_ = (expr); if (rc_) return rc_; }         \
    for (int k_ = 0; k_ < (e)->n_impl; k_++) { (e)->impl = (e)->impls[k_]; int rc_ = (expr); if (rc_) { (e)->impl = (e)->impls[0]; return rc_; } } \
    (e)->impl = (e)->impls[0]; } while (0)

void xck_destroy(xck_engine* e) {
    if (!e) return;
    for (int k = 0; k < e->n_impl; k++) { e->impl = e->impls[k]; engine_destroy(e); }
    delete e;
}
int xck_umi_bits(const xck_engine* e) { return e ? e->umi_bits : 0; }

int xck_push_batch(xck_engine* e, const xck_batch* b) { if (!e || !b) return XCK_E_ARG; FOR_IMPLS(e, engine_push(e, b, false)); return XCK_OK; }
int xck_push_batch_device(xck_engine* e, const xck_batch* b) { if (!e || !b) return XCK_E_ARG; FOR_IMPLS(e, engine_push(e, b, true)); return XCK_OK; }
int xck_flush(xck_engine* e) { if (!e) return XCK_E_ARG; FOR_IMPLS(e, engine_flush(e)); return XCK_OK; }
int xck_reset(xck_engine* e) { if (!e) return XCK_E_ARG; FOR_IMPLS(e, engine_reset(e)); return XCK_OK; }

int xck_finish_async(xck_engine* e) { if (!e) return XCK_E_ARG; FOR_IMPLS(e, engine_finish_async(e)); return XCK_OK; }

int xck_finish(xck_engine* e, xck_result* out) {
    if (!e || !out) return XCK_E_ARG;
    memset(out, 0, sizeof *out);
    if (e->n_impl == 0) { e->impl = nullptr; return engine_finish(e, out); }
    for (int k = 0; k < e->n_impl; k++) {
        xck_result r;
        e->impl = e->impls[k];
        int rc = engine_finish(e, &r);
        if (rc) { e->impl = e->impls[0]; return rc; }
        if (e->mode == XCK_MODE_BOTH) { if (k == 0) out->count = r.count; else { out->ad = r.ad; out->dp = r.dp; out->oth = r.oth; } }
        else *out = r;
    }
    e->impl = e->impls[0];
    return XCK_OK;
}

int xck_get_result_device(xck_engine* e, xck_result* out) {
    if (!e || !out) return XCK_E_ARG;
    memset(out, 0, sizeof *out);
    if (e->n_impl == 0) { e->impl = nullptr; return engine_result_device(e, out); }
    for (int k = 0; k < e->n_impl; k++) {
        xck_result r;
        e->impl = e->impls[k];
        int rc = engine_result_device(e, &r);
        if (rc) { e->impl = e->impls[0]; return rc; }
        if (e->mode == XCK_MODE_BOTH) { if (k == 0) out->count = r.count; else { out->ad = r.ad; out->dp = r.dp; out->oth = r.oth; } }
        else *out = r;
    }
    e->impl = e->impls[0];
    return XCK_OK;
}

int xck_get_stats(const xck_engine* ce, xck_stats* out) {
    if (!ce || !out) return XCK_E_ARG;
    xck_engine* e = const_cast<xck_engine*>(ce);
    if (e->n_impl == 0) return XCK_E_STATE;
    memset(out, 0, sizeof *out);
    for (int k = 0; k < e->n_impl; k++) {
        xck_stats s; e->impl = e->impls[k];
        int rc = engine_stats(e, &s);
        if (rc) { e->impl = e->impls[0]; return rc; }
        if (k == 0) *out = s;
        else { out->n_hits += s.n_hits; out->n_hits_unique += s.n_hits_unique; out->ms_h2d += s.ms_h2d; out->ms_device += s.ms_device;
               out->ms_join += s.ms_join; out->ms_sort += s.ms_sort; out->ms_d2h += s.ms_d2h; out->algo_bytes_join += s.algo_bytes_join; out->n_join_launches += s.n_join_launches; }
    }
    e->impl = e->impls[0];
    return XCK_OK;
}

// merge_mtx() header + body, rdr/fc/utils.py:54-93 (byte-identical text)
int xck_write_mtx(const char* path, const xck_coo* m, const int32_t* row_map, int32_t n_rows_out, int32_t n_cols) {
    if (!path || !m || !row_map) return XCK_E_ARG;
    FILE* fp = fopen(path, "wb");
    if (!fp) { set_thread_error(std::string("cannot open ") + path); return XCK_E_IO; }
    static const size_t BUF = 1 << 22;
    std::string buf; buf.reserve(BUF + 64);
    int64_t nnz = 0;
    for (int64_t i = 0; i < m->nnz; i++) if (row_map[m->row[i]] > 0) nnz++;
    char line[96];
    int k = snprintf(line, sizeof line, "%%%%MatrixMarket matrix coordinate integer general\n%%%%\n%d\t%d\t%lld\n", n_rows_out, n_cols, (long long)nnz);
    buf.append(line, (size_t)k);
    auto put_int = [&](int64_t v) { char t[24]; int n = 0; if (v == 0) t[n++] = '0'; bool neg = v < 0; if (neg) v = -v;
        while (v) { t[n++] = char('0' + v % 10); v /= 10; } if (neg) buf.push_back('-'); while (n) buf.push_back(t[--n]); };
    for (int64_t i = 0; i < m->nnz; i++) {
        int32_t r = row_map[m->row[i]];
        if (r <= 0) continue;
        put_int(r); buf.push_back('\t'); put_int((int64_t)m->col[i] + 1); buf.push_back('\t'); put_int(m->val[i]); buf.push_back('\n');
        if (buf.size() >= BUF) { if (fwrite(buf.data(), 1, buf.size(), fp) != buf.size()) { fclose(fp); return XCK_E_IO; } buf.clear(); }
    }
    if (!buf.empty() && fwrite(buf.data(), 1, buf.size(), fp) != buf.size()) { fclose(fp); return XCK_E_IO; }
    if (fclose(fp) != 0) return XCK_E_IO;
    return XCK_OK;
}

}  // extern "C"
