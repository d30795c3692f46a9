// host_model.hpp — host side of libecne_hip: the .r1cs reader, the trusted-function abstraction
// and the layout step that turns a constraint system into the flat arrays the HIP engine runs on.
//
// Mirrors, by behaviour (not by code), these reference functions:
//   readR1CS                   /root/reference/src/ParseR1CS.jl:50-124
//   abstraction                /root/reference/src/R1CSConstraintSolver.jl:237-395 (+ :205-235)
//   the trusted-function loop  :513-544 of solveWithTrustedFunctions
// Everything here is load-time data preparation; no propagation rule is evaluated on the host.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "fp256.hpp"
#include "jlorder.hpp"

namespace ecne {

enum Status : int {
    K_OK = 0,
    K_EFORMAT = -1,     // ParseR1CS.jl:58,62,69 asserts / truncated file
    K_EBOUNDS = -2,     // BoundsError: variable_states[-1] (:916), special input indexing (:762, :785)
    K_EDIVZERO = -3,    // DivideError from divexact by zero (:919-920, :1467)
    K_EUNDEF_DSU = -4,  // UndefVarError `dsu` (:762) when secp_solve == false
    K_EKEY = -5,        // KeyError in abstraction's variable map (:381-382)
    K_EDETSIZE = -6,    // linear-system group with more than 10 unknowns (reference: k!*k steps)
    K_EIO = -7,
    K_ENODEVICE = -8,   // no HIP device / HIP runtime error
    K_EINVAL = -9,
    K_ECAPACITY = -10,  // an internal device table overflowed (never silently truncated)
};

// One constraint system in "dictionary order": for every row part the entries appear in the
// order the reference's DefaultDict would iterate them, explicit zero coefficients included
// (an empty part is the single entry {1 => 0}, ParseR1CS.jl:113-115).
struct Rows {
    std::vector<uint64_t> ptr[3];     // size nC+1 each, offsets into var/coef
    std::vector<uint32_t> var[3];     // 1-based variable id (= wire id + 1)
    std::vector<fp::u256> coef[3];    // canonical residue
    size_t n() const { return ptr[0].empty() ? 0 : ptr[0].size() - 1; }
    void start() {
        for (int p = 0; p < 3; ++p) { ptr[p].assign(1, 0); var[p].clear(); coef[p].clear(); }
    }
    void append_row_from(const Rows& src, size_t i) {
        for (int p = 0; p < 3; ++p) {
            for (uint64_t k = src.ptr[p][i]; k < src.ptr[p][i + 1]; ++k) {
                var[p].push_back(src.var[p][k]);
                coef[p].push_back(src.coef[p][k]);
            }
            ptr[p].push_back(var[p].size());
        }
    }
};

struct R1CSFile {
    uint32_t field_size = 0, n_wires = 0, n_pub_out = 0, n_pub_in = 0, n_prv_in = 0, n_cons = 0;
    uint64_t n_labels = 0;
    Rows rows;
    std::vector<int64_t> knowns, outputs;
    int64_t n_vars = 0;
    uint64_t nnz[3] = {0, 0, 0};
    // file-order CSR views handed out by ecne_r1cs_csr (non-zero entries only): built on first use from
    // the file (the solve path never needs them)
    std::string path;
    bool csr_built = false;
    std::vector<uint64_t> csr_ptr[3];
    std::vector<uint32_t> csr_col[3];
    std::vector<uint64_t> csr_coef[3];
};

struct Special {
    std::string name;
    std::vector<int64_t> inputs, outputs;
};

// read-only view of a whole file (mmap; the 100+ MB constraint section is parsed in place)
struct FileView {
    const uint8_t* data = nullptr;
    size_t size = 0;
    bool ok = false;
    explicit FileView(const char* path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return;
        struct stat st;
        if (::fstat(fd, &st) == 0) {
            size = (size_t)st.st_size;
            if (size == 0) ok = true;
            else {
                void* m = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) { data = (const uint8_t*)m; ok = true; }
            }
        }
        ::close(fd);
    }
    ~FileView() { if (data) ::munmap((void*)data, size); }
    FileView(const FileView&) = delete;
    FileView& operator=(const FileView&) = delete;
};

inline uint32_t rd32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
inline uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

// readR1CS semantics: magic unchecked, version == 1, exactly 3 sections of
// type 1..3 in any order, prime never compared, coefficient width fixed at 32 bytes, duplicate
// wire ids "last wins" at the first occurrence's position, wire id == nWires accepted.
inline int load_r1cs(const char* path, R1CSFile& out) {
    FileView fv(path);
    if (!fv.ok) return K_EIO;
    const size_t N = fv.size;
    const uint8_t* b = fv.data;
    auto need = [&](size_t off, size_t len) { return off + len <= N; };
    if (!need(0, 12)) return K_EFORMAT;
    if (rd32(b + 4) != 1) return K_EFORMAT;
    if (rd32(b + 8) != 3) return K_EFORMAT;
    size_t cur = 12, start[4] = {0, 0, 0, 0};
    bool seen[4] = {false, false, false, false};
    for (int s = 0; s < 3; ++s) {
        if (!need(cur, 12)) return K_EFORMAT;
        uint32_t t = rd32(b + cur);
        if (t < 1 || t > 3) return K_EFORMAT;
        start[t] = cur + 12;
        seen[t] = true;
        cur += 12 + (size_t)rd64(b + cur + 4);
    }
    if (!seen[1] || !seen[2]) return K_EFORMAT;
    size_t h = start[1];
    if (!need(h, 4)) return K_EFORMAT;
    out.field_size = rd32(b + h);
    h += 4 + out.field_size;
    if (!need(h, 28)) return K_EFORMAT;
    out.n_wires = rd32(b + h);
    out.n_pub_out = rd32(b + h + 4);
    out.n_pub_in = rd32(b + h + 8);
    out.n_prv_in = rd32(b + h + 12);
    out.n_labels = rd64(b + h + 16);
    out.n_cons = rd32(b + h + 24);

    // pass 1 over the constraint section: term counts per part (bounds-checks the section, sizes the arrays)
    uint64_t terms[3] = {0, 0, 0};
    {
        size_t c = start[2];
        for (uint32_t r = 0; r < out.n_cons; ++r)
            for (int p = 0; p < 3; ++p) {
                if (!need(c, 4)) return K_EFORMAT;
                const uint32_t n = rd32(b + c);
                c += 4;
                if (!need(c, (size_t)n * 36)) return K_EFORMAT;
                c += (size_t)n * 36;
                terms[p] += n ? n : 1;   // an empty part is stored as {1 => 0}
            }
    }
    size_t c = start[2];
    out.rows.start();
    for (int p = 0; p < 3; ++p) {
        out.rows.ptr[p].reserve((size_t)out.n_cons + 1);
        out.rows.var[p].reserve(terms[p]);
        out.rows.coef[p].reserve(terms[p]);
    }
    out.path = path;
    out.csr_built = false;
    jl::SlotTable tab;
    std::vector<fp::u256> tmpc;
    for (uint32_t r = 0; r < out.n_cons; ++r) {
        for (int p = 0; p < 3; ++p) {
            uint32_t n = rd32(b + c);
            c += 4;
            if (n == 0) {
                out.rows.var[p].push_back(1);
                out.rows.coef[p].push_back(fp::make(0));
            } else if (n == 1) {   // one term: no dictionary order to reproduce
                const uint32_t wire = rd32(b + c);
                const fp::u256 v = fp::reduce(fp::make(rd64(b + c + 4), rd64(b + c + 12), rd64(b + c + 20), rd64(b + c + 28)));
                c += 36;
                out.rows.var[p].push_back(wire + 1);
                out.rows.coef[p].push_back(v);
                if (!fp::is_zero(v)) out.nnz[p]++;
            } else {
                tab.reset();
                tmpc.clear();
                for (uint32_t k = 0; k < n; ++k) {
                    uint32_t wire = rd32(b + c);
                    fp::u256 v = fp::make(rd64(b + c + 4), rd64(b + c + 12), rd64(b + c + 20), rd64(b + c + 28));
                    c += 36;
                    v = fp::reduce(v);
                    bool ins;
                    int64_t& slot = tab.upsert((int64_t)wire + 1, (int64_t)tmpc.size(), ins);
                    if (ins) tmpc.push_back(v); else tmpc[(size_t)slot] = v;
                }
                tab.for_each([&](int64_t key, int64_t pay) {
                    out.rows.var[p].push_back((uint32_t)key);
                    out.rows.coef[p].push_back(tmpc[(size_t)pay]);
                    if (!fp::is_zero(tmpc[(size_t)pay])) out.nnz[p]++;
                });
            }
            out.rows.ptr[p].push_back(out.rows.var[p].size());
        }
    }
    out.knowns.assign(1, 1);
    for (int64_t i = 2 + (int64_t)out.n_pub_out; i <= 1 + (int64_t)out.n_pub_out + out.n_pub_in + out.n_prv_in; ++i)
        out.knowns.push_back(i);
    out.outputs.clear();
    for (int64_t i = 2; i <= 1 + (int64_t)out.n_pub_out; ++i) out.outputs.push_back(i);
    out.n_vars = (int64_t)out.n_wires + 1;
    return K_OK;
}

// file-order CSR of a loaded file (what ecne_r1cs_csr hands out): every non-zero term as stored, duplicates
// included, coefficients reduced. Built on first use by reading the file again.
inline int build_file_csr(R1CSFile& out) {
    if (out.csr_built) return K_OK;
    FileView fv(out.path.c_str());
    if (!fv.ok) return K_EIO;
    const size_t N = fv.size;
    const uint8_t* b = fv.data;
    auto need = [&](size_t off, size_t len) { return off + len <= N; };
    if (!need(0, 12)) return K_EFORMAT;
    size_t cur = 12, start2 = 0;
    for (int s = 0; s < 3; ++s) {
        if (!need(cur, 12)) return K_EFORMAT;
        if (rd32(b + cur) == 2) start2 = cur + 12;
        cur += 12 + (size_t)rd64(b + cur + 4);
    }
    if (!start2) return K_EFORMAT;
    size_t c = start2;
    for (int p = 0; p < 3; ++p) {
        out.csr_ptr[p].assign(1, 0);
        out.csr_col[p].clear();
        out.csr_coef[p].clear();
    }
    for (uint32_t r = 0; r < out.n_cons; ++r)
        for (int p = 0; p < 3; ++p) {
            if (!need(c, 4)) return K_EFORMAT;
            const uint32_t n = rd32(b + c);
            c += 4;
            if (!need(c, (size_t)n * 36)) return K_EFORMAT;
            for (uint32_t k = 0; k < n; ++k) {
                const uint32_t wire = rd32(b + c);
                const fp::u256 v = fp::reduce(fp::make(rd64(b + c + 4), rd64(b + c + 12), rd64(b + c + 20), rd64(b + c + 28)));
                c += 36;
                if (!fp::is_zero(v)) {
                    out.csr_col[p].push_back(wire + 1);
                    for (int w = 0; w < 4; ++w) out.csr_coef[p].push_back(v.w[w]);
                }
            }
            out.csr_ptr[p].push_back(out.csr_col[p].size());
        }
    out.csr_built = true;
    return K_OK;
}

// ------------------------------------------------------------------ abstraction (§8f-1, host side)
// A trusted sub-circuit occurrence = a window of consecutive rows whose per-part multisets of
// non-zero coefficients equal the sub-circuit's, and whose variables can be put in bijection with
// the sub-circuit's variables by sorting both sides by "appearance signature" (which parts of
// which rows, with which coefficient).  Ties in that sort are broken by the reference's hash-table
// order, which is why the appearance maps are SlotTables filled in dictionary order.
namespace detail {
struct LessU256 {
    bool operator()(const fp::u256& a, const fp::u256& b) const { return fp::cmp(a, b) < 0; }
};
inline uint64_t mix64(uint64_t h, uint64_t x) {
    h ^= x + 0x9e3779b97f4a7c15ULL + (h << 6) + (h >> 2);
    h *= 0xff51afd7ed558ccdULL;
    return h ^ (h >> 32);
}
// sorted non-zero coefficients of one part
inline void part_values(const Rows& R, int p, size_t i, std::vector<fp::u256>& out) {
    out.clear();
    for (uint64_t k = R.ptr[p][i]; k < R.ptr[p][i + 1]; ++k)
        if (!fp::is_zero(R.coef[p][k])) out.push_back(R.coef[p][k]);
    if (out.size() > 1) std::sort(out.begin(), out.end(), LessU256());
}
inline uint64_t row_fingerprint(const Rows& R, size_t i, std::vector<fp::u256>& v) {   // v: scratch
    uint64_t h = 0x1234567;
    for (int p = 0; p < 3; ++p) {
        // (most parts hold at most one non-zero value: nothing to sort, nothing to allocate)
        const uint64_t k0 = R.ptr[p][i], k1 = R.ptr[p][i + 1];
        uint64_t nzc = 0, last = 0;
        for (uint64_t k = k0; k < k1 && nzc < 2; ++k)
            if (!fp::is_zero(R.coef[p][k])) { ++nzc; last = k; }
        if (nzc == 0) continue;
        if (nzc == 1) { for (int w = 0; w < 4; ++w) h = mix64(h, R.coef[p][last].w[w]); continue; }
        part_values(R, p, i, v);
        for (auto& x : v)
            for (int w = 0; w < 4; ++w) h = mix64(h, x.w[w]);
    }
    return h;  // parts are NOT delimited: the reference hashes the concatenation (:231-232)
}
typedef std::vector<std::pair<int64_t, fp::u256>> Appear;
inline bool appear_less(const Appear& x, const Appear& y) {
    size_t n = std::min(x.size(), y.size());
    for (size_t i = 0; i < n; ++i) {
        if (x[i].first != y[i].first) return x[i].first < y[i].first;
        int c = fp::cmp(x[i].second, y[i].second);
        if (c) return c < 0;
    }
    return x.size() < y.size();
}
inline bool appear_eq(const Appear& x, const Appear& y) {
    if (x.size() != y.size()) return false;
    for (size_t i = 0; i < x.size(); ++i)
        if (x[i].first != y[i].first || !fp::eq(x[i].second, y[i].second)) return false;
    return true;
}
struct AppearMap {
    jl::SlotTable tab;
    std::vector<Appear> lists;   // lists[0 .. used): one per variable; the vectors are reused across clear()
    size_t used = 0;
    void clear() { tab.reset(); used = 0; }
    void add(int64_t var, int64_t where, const fp::u256& c) {
        bool ins;
        int64_t& s = tab.upsert(var, (int64_t)used, ins);
        if (ins) {
            if (used == lists.size()) lists.emplace_back(); else lists[used].clear();
            ++used;
        }
        lists[(size_t)s].push_back({where, c});
    }
    // (variable, list index) sorted by list, stable w.r.t. table order
    std::vector<std::pair<int64_t, int64_t>> sorted() const {
        std::vector<std::pair<int64_t, int64_t>> v;
        tab.for_each([&](int64_t key, int64_t pay) { v.push_back({key, pay}); });
        std::stable_sort(v.begin(), v.end(), [&](const std::pair<int64_t, int64_t>& a, const std::pair<int64_t, int64_t>& b) {
            return appear_less(lists[(size_t)a.second], lists[(size_t)b.second]);
        });
        return v;
    }
};
}  // namespace detail

// Replaces every (greedy, left to right, stuck-cursor) occurrence of `sub` in `rows` by a special
// constraint; returns K_OK or K_EKEY.
inline int abstract_one(const std::string& name, Rows& rows, const R1CSFile& sub, std::vector<Special>& specials) {
    using namespace detail;
    const size_t nC = rows.n(), nS = sub.rows.n();
    std::vector<size_t> cand;
    if (nC + 1 >= nS + 1 && nC >= nS) {
        std::vector<uint64_t> fb(nC), fs(nS);
        std::vector<fp::u256> scratch;
        for (size_t i = 0; i < nC; ++i) fb[i] = row_fingerprint(rows, i, scratch);
        for (size_t i = 0; i < nS; ++i) fs[i] = row_fingerprint(sub.rows, i, scratch);
        for (size_t i = 0; i + nS <= nC; ++i) {
            bool m = true;
            for (size_t j = 0; j + 1 < nS; ++j)
                if (fb[i + j] != fs[j]) { m = false; break; }
            if (m) cand.push_back(i);
        }
    }
    AppearMap orig;
    {
        int64_t counter = 1;
        for (size_t j = 0; j < nS; ++j)
            for (int p = 0; p < 3; ++p) {
                for (uint64_t k = sub.rows.ptr[p][j]; k < sub.rows.ptr[p][j + 1]; ++k)
                    if (!fp::is_zero(sub.rows.coef[p][k])) orig.add(sub.rows.var[p][k], counter, sub.rows.coef[p][k]);
                ++counter;
            }
    }
    const auto l2 = orig.sorted();
    struct Match { size_t at; std::unordered_map<int64_t, int64_t> map; };
    std::vector<Match> matches;
    std::vector<fp::u256> va, vb;
    // the pattern's sorted coefficient lists, once (every candidate window is compared against them)
    std::vector<fp::u256> subvals;
    std::vector<size_t> subptr(1, 0);
    if (!cand.empty())
        for (size_t j = 0; j < nS; ++j)
            for (int p = 0; p < 3; ++p) {
                part_values(sub.rows, p, j, vb);
                subvals.insert(subvals.end(), vb.begin(), vb.end());
                subptr.push_back(subvals.size());
            }
    AppearMap cur;
    for (size_t at : cand) {
        cur.clear();
        int64_t counter = 0;
        bool ok = true;
        for (size_t j = 0; j < nS && ok; ++j)
            for (int p = 0; p < 3 && ok; ++p) {
                ++counter;
                part_values(rows, p, at + j, va);
                const size_t s0 = subptr[j * 3 + (size_t)p], s1 = subptr[j * 3 + (size_t)p + 1];
                if (va.size() != s1 - s0) { ok = false; break; }
                for (size_t t = 0; t < va.size(); ++t)
                    if (!fp::eq(va[t], subvals[s0 + t])) { ok = false; break; }
                if (!ok) break;
                for (uint64_t k = rows.ptr[p][at + j]; k < rows.ptr[p][at + j + 1]; ++k)
                    if (!fp::is_zero(rows.coef[p][k])) cur.add(rows.var[p][k], counter, rows.coef[p][k]);
            }
        if (!ok) continue;
        const auto l1 = cur.sorted();
        if (l1.size() != l2.size()) continue;
        for (size_t x = 0; x < l1.size() && ok; ++x)
            if (!appear_eq(cur.lists[(size_t)l1[x].second], orig.lists[(size_t)l2[x].second])) ok = false;
        if (!ok) continue;
        Match m;
        m.at = at;
        for (size_t x = 0; x < l1.size(); ++x) m.map[l2[x].first] = l1[x].first;
        matches.push_back(std::move(m));
    }
    Rows red;
    red.start();
    for (int p = 0; p < 3; ++p) {
        red.ptr[p].reserve(rows.ptr[p].size());
        red.var[p].reserve(rows.var[p].size());
        red.coef[p].reserve(rows.coef[p].size());
    }
    // rows outside the matched windows are copied range by range
    auto copy_range = [&](size_t a, size_t b) {   // rows [a, b)
        for (int p = 0; p < 3; ++p) {
            const uint64_t k0 = rows.ptr[p][a], k1 = rows.ptr[p][b];
            const uint64_t shift = (uint64_t)red.var[p].size() - k0;
            red.var[p].insert(red.var[p].end(), rows.var[p].begin() + (ptrdiff_t)k0, rows.var[p].begin() + (ptrdiff_t)k1);
            red.coef[p].insert(red.coef[p].end(), rows.coef[p].begin() + (ptrdiff_t)k0, rows.coef[p].begin() + (ptrdiff_t)k1);
            for (size_t r = a + 1; r <= b; ++r) red.ptr[p].push_back(rows.ptr[p][r] + shift);
        }
    };
    size_t i = 0;
    for (size_t mi = 0; mi <= matches.size(); ++mi) {
        // greedy left to right with the reference's stuck cursor (:368-388): a match that starts inside the
        // previous window is never reached again, and neither is any later one
        const size_t stop = mi < matches.size() ? matches[mi].at : nC;
        if (stop < i) { copy_range(i, nC); i = nC; break; }
        copy_range(i, stop);
        i = stop;
        if (mi == matches.size()) break;
        Special sp;
        sp.name = name;
        for (int64_t x : sub.knowns)
            if (x != 1) {
                auto it = matches[mi].map.find(x);
                if (it == matches[mi].map.end()) return K_EKEY;
                sp.inputs.push_back(it->second);
            }
        for (int64_t x : sub.outputs) {
            auto it = matches[mi].map.find(x);
            if (it == matches[mi].map.end()) return K_EKEY;
            sp.outputs.push_back(it->second);
        }
        specials.push_back(std::move(sp));
        i += nS;
    }
    rows = std::move(red);
    return K_OK;
}

}  // namespace ecne
