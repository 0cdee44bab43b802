// snptext.cpp - host-side fast path of the phased-SNP loaders (a 1 M-SNP list costs the Python loaders ~1 s, a third of a
// 100 M-read `baf` run).  Same accept / reject decisions as xcltk_amd/fc_common.py load_snp_from_tsv / load_snp_from_vcf,
// i.e. as the reference's baf/fc/utils.py:51-110 and :114-193, for the files this parser declares itself eligible for:
// pure ASCII without carriage returns and with plain decimal positions.  Anything else returns 1 and the caller uses the
// generic Python loader, so behaviour (including its exceptions) is unchanged there.
#include <zlib.h>
#include <cstdint>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include "xck_internal.h"

namespace {

struct SnpTextImpl {
    xck_snp_text pub;
    std::vector<int32_t> chrom_id; std::vector<int64_t> pos; std::string ref, alt; std::vector<int8_t> rh, ah;
    std::vector<std::string> chrom_store; std::vector<const char*> chrom_ptr;
    std::vector<int64_t> rej_line; std::vector<int8_t> rej_code;
};

inline bool py_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }   // str.isspace() over ASCII
inline int base_code(unsigned char c) { if (c >= 'a' && c <= 'z') c -= 32; return (c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N') ? c : 0; }

struct Field { const char* p; size_t n; };

// line.rstrip().split("\t")
inline void split_tabs(const char* s, size_t n, std::vector<Field>& out) {
    while (n > 0 && py_space((unsigned char)s[n - 1])) n--;
    out.clear();
    size_t a = 0;
    for (size_t i = 0; i <= n; i++) if (i == n || s[i] == '\t') { out.push_back({s + a, i - a}); a = i + 1; }
}
inline void split_char(const Field& f, char sep, std::vector<Field>& out) {
    out.clear();
    size_t a = 0;
    for (size_t i = 0; i <= f.n; i++) if (i == f.n || f.p[i] == sep) { out.push_back({f.p + a, i - a}); a = i + 1; }
}
inline bool is_str(const Field& f, const char* lit) { const size_t l = strlen(lit); return f.n == l && memcmp(f.p, lit, l) == 0; }

// int(text) for the plain form only; anything Python might still accept (sign, blanks, underscores) makes the file ineligible
inline bool plain_int(const Field& f, int64_t& v) {
    if (f.n == 0 || f.n > 18) return false;
    v = 0;
    for (size_t i = 0; i < f.n; i++) { if (f.p[i] < '0' || f.p[i] > '9') return false; v = v * 10 + (f.p[i] - '0'); }
    return true;
}

}  // namespace

extern "C" {

static int parse_snp_text_impl(const char* path, int is_vcf, xck_snp_text** out) {
    if (!path || !out) return XCK_E_ARG;
    *out = nullptr;
    gzFile gz = gzopen(path, "rb");                                         // plain files are read through unchanged
    if (!gz) { xck::set_thread_error(std::string("cannot open ") + path); return XCK_E_IO; }
    gzbuffer(gz, 1 << 20);
    std::string data;
    { std::vector<char> buf(1 << 22); int k;
      while ((k = gzread(gz, buf.data(), (unsigned)buf.size())) > 0) data.append(buf.data(), (size_t)k);
      const bool bad = k < 0; gzclose(gz);
      if (bad) { xck::set_thread_error(std::string("read error in ") + path); return XCK_E_IO; } }
    for (unsigned char c : data) if (c >= 0x80 || c == '\r' || c == 0) return 1;      // decoding / universal newlines: the generic loader's business
    SnpTextImpl* t = new SnpTextImpl();
    std::unordered_map<std::string, int32_t> chrom_idx;
    std::vector<Field> parts, fields, values, gts;
    const char* s = data.data(); const size_t total = data.size();
    size_t at = 0; int64_t nl = 0;
    bool eligible = true;
    while (at < total && eligible) {
        const char* e = (const char*)memchr(s + at, '\n', total - at);
        const size_t len = e ? (size_t)(e - (s + at)) : total - at;
        const char* line = s + at;
        at += len + (e ? 1 : 0);
        nl++;
        Field chrom, posf, a1, a2; int rb, ab;
        auto reject = [&](int code) { t->rej_line.push_back(nl); t->rej_code.push_back((int8_t)code); };
        if (!is_vcf) {
            if (nl == 1) continue;                                          // header
            split_tabs(line, len, parts);
            if (parts.size() < 6) { reject(XCK_SNP_REJ_COLUMNS); continue; }
            chrom = parts[0]; posf = parts[1];
            rb = parts[2].n == 1 ? base_code((unsigned char)parts[2].p[0]) : 0;
            ab = parts[3].n == 1 ? base_code((unsigned char)parts[3].p[0]) : 0;
            a1 = parts[4]; a2 = parts[5];
        } else {
            if (len == 0 || line[0] == '#') continue;                       // line[0] in ("#", "\n")
            split_tabs(line, len, parts);
            if (parts.size() < 10) { reject(XCK_SNP_REJ_COLUMNS); continue; }
            chrom = parts[0]; posf = parts[1];
            rb = parts[3].n == 1 ? base_code((unsigned char)parts[3].p[0]) : 0;
            ab = parts[4].n == 1 ? base_code((unsigned char)parts[4].p[0]) : 0;
            if (!rb) { reject(XCK_SNP_REJ_REF); continue; }
            if (!ab) { reject(XCK_SNP_REJ_ALT); continue; }
            split_char(parts[8], ':', fields); split_char(parts[9], ':', values);
            size_t gi = fields.size();
            for (size_t i = 0; i < fields.size(); i++) if (is_str(fields[i], "GT")) { gi = i; break; }
            if (gi == fields.size()) { reject(XCK_SNP_REJ_NO_GT); continue; }
            if (values.size() != fields.size()) { reject(XCK_SNP_REJ_FORMAT_LEN); continue; }
            const Field gt = values[gi];
            char sep = 0;
            if (memchr(gt.p, '|', gt.n)) sep = '|'; else if (memchr(gt.p, '/', gt.n)) sep = '/';
            if (!sep) { reject(XCK_SNP_REJ_DELIMITER); continue; }
            split_char(gt, sep, gts);                                       // >= 2 pieces because sep occurs
            a1 = gts[0]; a2 = gts[1];
        }
        if (!rb) { reject(XCK_SNP_REJ_REF); continue; }
        if (!ab) { reject(XCK_SNP_REJ_ALT); continue; }
        const bool g01 = is_str(a1, "0") && is_str(a2, "1"), g10 = is_str(a1, "1") && is_str(a2, "0");
        if (!g01 && !g10) { reject(XCK_SNP_REJ_GT); continue; }
        int64_t pv;
        if (!plain_int(posf, pv)) { eligible = false; break; }              // int() of an accepted line: leave every odd spelling to Python
        size_t skip = (chrom.n >= 3 && (chrom.p[0] | 32) == 'c' && (chrom.p[1] | 32) == 'h' && (chrom.p[2] | 32) == 'r') ? 3 : 0;
        std::string cname(chrom.p + skip, chrom.n - skip);
        auto it = chrom_idx.find(cname);
        int32_t ci;
        if (it == chrom_idx.end()) { ci = (int32_t)t->chrom_store.size(); chrom_idx.emplace(cname, ci); t->chrom_store.push_back(cname); } else ci = it->second;
        t->chrom_id.push_back(ci); t->pos.push_back(pv); t->ref.push_back((char)rb); t->alt.push_back((char)ab);
        t->rh.push_back(g01 ? 0 : 1); t->ah.push_back(g01 ? 1 : 0);
    }
    if (!eligible) { delete t; return 1; }
    for (auto& c : t->chrom_store) t->chrom_ptr.push_back(c.c_str());
    t->pub.n = (int64_t)t->pos.size();
    t->pub.chrom_id = t->chrom_id.data(); t->pub.pos = t->pos.data(); t->pub.ref = t->ref.data(); t->pub.alt = t->alt.data();
    t->pub.ref_hap = t->rh.data(); t->pub.alt_hap = t->ah.data();
    t->pub.n_chroms = (int32_t)t->chrom_ptr.size(); t->pub.chroms = t->chrom_ptr.data();
    t->pub.n_rejected = (int64_t)t->rej_line.size(); t->pub.rej_line = t->rej_line.data(); t->pub.rej_code = t->rej_code.data();
    *out = &t->pub;
    return XCK_OK;
}

int xck_parse_snp_text(const char* path, int is_vcf, xck_snp_text** out) {
    try { return parse_snp_text_impl(path, is_vcf, out); }                   // no C++ exception crosses the C ABI
    catch (const std::bad_alloc&) { xck::set_thread_error("out of host memory"); return XCK_E_NOMEM; }
    catch (const std::exception& x) { xck::set_thread_error(x.what()); return XCK_E_IO; }
    catch (...) { xck::set_thread_error("unknown C++ exception"); return XCK_E_IO; }
}

void xck_free_snp_text(xck_snp_text* t) { delete reinterpret_cast<SnpTextImpl*>(t); }   // pub is the first member

}  // extern "C"
