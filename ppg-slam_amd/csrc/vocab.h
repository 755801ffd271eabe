// DBoW3 vocabulary file reader (host).  Pure C++ (no HIP): included by ppg_front.cpp.
//
// DBoW3 is a third-party dependency of the reference that is not vendored (CMakeLists.txt:16-17, README.md:30); the
// vocabularies the reference ships (Vocabulary/voc_euroc_9x3.gz, voc_tum_9x3.gz: binary files despite the suffix) are
// what DBoW3::Vocabulary::toStream writes:
//   u64 signature 88877711233, u8 compressed, u32 node count, then (compressed) u32 chunk count and that many QuickLZ
//   packets of at most 10000 payload bytes, else the payload directly.  Payload: i32 k, i32 L, i32 scoring, i32 weighting;
//   per node except the root, in creation order: u32 id, u32 parent, f64 weight, i32 cols, i32 rows, i32 type, cols x
//   rows elements (CV_32F = 5 here); u32 word count and (u32 word id, u32 node id) per word.
// QuickLZ 1.5 packets, level 1, no streaming: header byte (bit 0 compressed, bit 1 four-byte sizes, bits 2-3 level)
// + compressed and decompressed sizes; a packet whose payload did not shrink is stored verbatim.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

struct ppg_vocab {
    unsigned long long serial = 0;    // identifies this object to a context's device copy
    int k = 0, L = 0, scoring = 0, weighting = 0;
    int n_nodes = 0, n_words = 0, dim = 0;
    std::vector<int32_t> parent;      // [n_nodes], root: -1
    std::vector<double> weight;       // [n_nodes]
    std::vector<int32_t> word_id;     // [n_nodes], -1 for inner nodes
    std::vector<int32_t> child_off;   // [n_nodes + 1] children in the order the file lists them (the order transform() visits)
    std::vector<int32_t> child_idx;
    std::vector<int32_t> word_node;   // [n_words]
    std::vector<float> desc;          // [n_nodes][dim], root row zero
};

namespace {

inline uint32_t vrd(const uint8_t* p, int n) {
    uint32_t v = 0;
    for (int i = 0; i < n; i++) v |= (uint32_t)p[i] << (8 * i);
    return v;
}

// One QuickLZ 1.5 packet (level 1).  Returns false on a malformed packet.  `src` must have 4 readable bytes past the
// packet (the decoder looks ahead), which the caller guarantees by padding its buffer.
bool qlz_packet(const uint8_t* src, size_t avail, std::vector<uint8_t>& out, size_t* consumed) {
    if (avail < 3) return false;
    const uint8_t fl = src[0];
    const int hdr = (fl & 2) ? 9 : 3;
    if (avail < (size_t)hdr) return false;
    const size_t cs = (fl & 2) ? vrd(src + 1, 4) : src[1], ds = (fl & 2) ? vrd(src + 5, 4) : src[2];
    if (cs < (size_t)hdr || cs > avail || ds > (1u << 24)) return false;
    *consumed = cs;
    const size_t base = out.size();
    out.resize(base + ds);
    uint8_t* dst = out.data() + base;
    if (!(fl & 1)) {   // stored
        if (cs != hdr + ds) return false;
        memcpy(dst, src + hdr, ds);
        return true;
    }
    if (((fl >> 2) & 3) != 1) return false;   // only level 1 is read (what DBoW3 builds QuickLZ with)
    if (ds == 0) return true;
    std::vector<int64_t> hash(4096, 0);
    const uint8_t* s = src + hdr;
    const uint8_t* send = src + cs;
    int64_t d = 0, last_hashed = -1;
    const int64_t last = (int64_t)ds - 1, last_matchstart = last - 6 - 4;
    uint32_t cword = 1;
    static const int bitlut[16] = {4, 0, 1, 0, 2, 0, 1, 0, 3, 0, 1, 0, 2, 0, 1, 0};
    auto rd3 = [&](int64_t p) {   // three output bytes at p, zero beyond the output
        uint32_t v = 0;
        for (int i = 0; i < 3; i++)
            if (p + i <= last) v |= (uint32_t)dst[p + i] << (8 * i);
        return v;
    };
    auto hash_upto = [&](int64_t upto) {
        while (last_hashed < upto) {
            last_hashed++;
            const uint32_t f = rd3(last_hashed);
            hash[((f >> 12) ^ f) & 4095u] = last_hashed;
        }
    };
    for (;;) {
        if (cword == 1) {
            if (s + 4 > send) return false;
            cword = vrd(s, 4);
            s += 4;
        }
        if (s > send) return false;
        const uint32_t fetch = vrd(s, 4);
        if (cword & 1u) {
            cword >>= 1;
            const int64_t off = hash[(fetch >> 4) & 0xfffu];
            int64_t ml;
            if (fetch & 0xfu) { ml = (fetch & 0xfu) + 2; s += 2; }
            else { ml = s[2]; s += 3; }
            if (off >= d || d + ml > (int64_t)ds) return false;
            for (int64_t i = 0; i < ml; i++) dst[d + i] = dst[off + i];   // may overlap forwards
            d += ml;
            hash_upto(d - ml);
            last_hashed = d - 1;
        } else if (d < last_matchstart) {
            const int n = bitlut[cword & 0xfu];
            for (int i = 0; i < n; i++) dst[d + i] = s[i];
            cword >>= n;
            d += n;
            s += n;
            hash_upto(d - 3);
        } else {
            while (d <= last) {
                if (cword == 1) { s += 4; cword = 1u << 31; }
                if (s >= send) return false;
                dst[d++] = *s++;
                cword >>= 1;
            }
            return true;
        }
    }
}

ppg_vocab* vocab_parse(const char* path, std::string& why) {
    FILE* f = fopen(path, "rb");
    if (!f) { why = std::string("cannot open ") + path; return nullptr; }
    std::vector<uint8_t> raw;
    uint8_t buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0) raw.insert(raw.end(), buf, buf + got);
    fclose(f);
    const size_t size = raw.size();
    raw.resize(size + 8, 0);   // look-ahead room for the packet decoder
    auto bad = [&](const char* what) { why = std::string(path) + ": " + what; return (ppg_vocab*)nullptr; };
    if (size < 13) return bad("too short for a DBoW3 binary vocabulary");
    uint64_t sig;
    memcpy(&sig, raw.data(), 8);
    if (sig != 88877711233ull) return bad("not a DBoW3 binary vocabulary (signature)");
    const bool compressed = raw[8] != 0;
    const uint32_t nnodes = vrd(raw.data() + 9, 4);
    if (nnodes < 2 || nnodes > (1u << 22)) return bad("implausible node count");
    std::vector<uint8_t> pay;
    if (compressed) {
        if (size < 17) return bad("truncated");
        const uint32_t nch = vrd(raw.data() + 13, 4);
        size_t o = 17;
        for (uint32_t c = 0; c < nch; c++) {
            size_t used = 0;
            if (o >= size || !qlz_packet(raw.data() + o, size - o, pay, &used)) return bad("malformed QuickLZ packet");
            o += used;
        }
    } else {
        pay.assign(raw.begin() + 13, raw.begin() + size);
    }
    size_t p = 0;
    auto need = [&](size_t n) { return p + n <= pay.size(); };
    if (!need(16)) return bad("truncated payload");
    ppg_vocab* v = new ppg_vocab();
    v->k = (int)vrd(pay.data(), 4); v->L = (int)vrd(pay.data() + 4, 4);
    v->scoring = (int)vrd(pay.data() + 8, 4); v->weighting = (int)vrd(pay.data() + 12, 4);
    p = 16;
    v->n_nodes = (int)nnodes;
    v->parent.assign(nnodes, -1);
    v->weight.assign(nnodes, 0.0);
    v->word_id.assign(nnodes, -1);
    std::vector<std::vector<int32_t>> children(nnodes);
    std::vector<uint8_t> seen(nnodes, 0);
    auto fail = [&](const char* what) { delete v; return bad(what); };
    if (v->k < 1 || v->k > 64 || v->L < 1 || v->L > 16 || v->scoring < 0 || v->scoring > 5 || v->weighting < 0 || v->weighting > 3)
        return fail("implausible k / L / scoring / weighting");
    for (uint32_t i = 1; i < nnodes; i++) {
        if (!need(28)) return fail("truncated node");
        const uint32_t id = vrd(pay.data() + p, 4), par = vrd(pay.data() + p + 4, 4);
        double w;
        memcpy(&w, pay.data() + p + 8, 8);
        const int cols = (int)vrd(pay.data() + p + 16, 4), rows = (int)vrd(pay.data() + p + 20, 4), type = (int)vrd(pay.data() + p + 24, 4);
        p += 28;
        if (id == 0 || id >= nnodes || par >= nnodes || seen[id]) return fail("bad node id");
        if (type != 5 || rows != 1 || cols < 1 || cols > 4096) return fail("node descriptors must be 1 x n CV_32F");
        if (v->dim == 0) { v->dim = cols; v->desc.assign((size_t)nnodes * cols, 0.f); }
        if (cols != v->dim) return fail("node descriptors of different lengths");
        if (!need((size_t)cols * 4)) return fail("truncated descriptor");
        memcpy(v->desc.data() + (size_t)id * cols, pay.data() + p, (size_t)cols * 4);
        p += (size_t)cols * 4;
        seen[id] = 1;
        v->parent[id] = (int32_t)par;
        v->weight[id] = w;
        children[par].push_back((int32_t)id);
    }
    if (!need(4)) return fail("truncated word table");
    const uint32_t nw = vrd(pay.data() + p, 4);
    p += 4;
    if (nw > nnodes || !need((size_t)nw * 8)) return fail("bad word table");
    v->n_words = (int)nw;
    v->word_node.assign(nw, -1);
    for (uint32_t i = 0; i < nw; i++) {
        const uint32_t wid = vrd(pay.data() + p, 4), nid = vrd(pay.data() + p + 4, 4);
        p += 8;
        if (wid >= nw || nid >= nnodes) return fail("bad word entry");
        v->word_id[nid] = (int32_t)wid;
        v->word_node[wid] = (int32_t)nid;
    }
    v->child_off.assign(nnodes + 1, 0);
    for (uint32_t i = 0; i < nnodes; i++) {
        v->child_off[i + 1] = v->child_off[i] + (int32_t)children[i].size();
        v->child_idx.insert(v->child_idx.end(), children[i].begin(), children[i].end());
    }
    // every leaf must be a word (transform() ends at a leaf and reads its word id), and the descent must terminate
    for (uint32_t i = 0; i < nnodes; i++)
        if (children[i].empty() && v->word_id[i] < 0 && i != 0) return fail("a leaf node is not a word");
    if (children[0].empty()) return fail("the root has no children");
    return v;
}

}  // namespace
