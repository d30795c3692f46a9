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
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <map>
#include <memory>
#include <string>
#include <thread>
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
    K_ETIMEOUT = -11,   // the workgroups of a job did not meet at their barrier in time (the device is shared)
    K_ENOCONVERGE = -12, // pop_cap reached: the reference's queue never drains on this input (it would not terminate)
};

// Host-side worker threads (parse, abstraction, layout): opt-in. The library works on the calling thread unless the
// caller asks for workers (ecne_set_host_threads, or ECNE_HOST_THREADS in the environment); at most 32.
// Results never depend on the thread count: every worker fills a range whose position was fixed beforehand.
inline std::atomic<unsigned>& host_threads_setting() {
    static std::atomic<unsigned> n{[] {
        const char* e = std::getenv("ECNE_HOST_THREADS");
        long h = e ? std::atol(e) : 1;
        if (e && h <= 0) h = (long)std::thread::hardware_concurrency();
        return (unsigned)std::min<long>(std::max<long>(h, 1), 32);
    }()};
    return n;
}
inline unsigned host_threads() { return host_threads_setting().load(std::memory_order_relaxed); }
inline unsigned set_host_threads(int n) {
    long h = n <= 0 ? (long)std::thread::hardware_concurrency() : n;
    host_threads_setting().store((unsigned)std::min<long>(std::max<long>(h, 1), 32), std::memory_order_relaxed);
    return host_threads();
}
// f(chunk, worker) for every chunk in [0, n_chunks), chunks handed out dynamically
// (a for_chunks inside a worker of another one runs on that worker: the parts of a split file are laid out side by side, each layout
//  serially -- not 32 x 32 threads)
inline bool& in_for_chunks_worker() { static thread_local bool b = false; return b; }
template <class F>
inline void for_chunks(size_t n_chunks, F&& f) {
    const unsigned T = in_for_chunks_worker() ? 1u : (unsigned)std::min<size_t>(host_threads(), n_chunks);
    if (T <= 1) {
        for (size_t c = 0; c < n_chunks; ++c) f(c, 0u);
        return;
    }
    std::atomic<size_t> next{0};
    auto work = [&](unsigned w) {
        const bool was = in_for_chunks_worker();
        in_for_chunks_worker() = true;
        for (size_t c; (c = next.fetch_add(1, std::memory_order_relaxed)) < n_chunks;) f(c, w);
        in_for_chunks_worker() = was;
    };
    std::vector<std::thread> pool;
    pool.reserve(T - 1);
    for (unsigned w = 1; w < T; ++w) pool.emplace_back(work, w);
    work(0);
    for (auto& t : pool) t.join();
}
inline unsigned for_chunks_workers(size_t n_chunks) { return in_for_chunks_worker() ? 1u : (unsigned)std::max<size_t>(1, std::min<size_t>(host_threads(), n_chunks)); }

// std::allocator whose resize() leaves trivially-constructible elements uninitialised: the big row arrays are
// sized once and then written (first touched) by the worker threads
template <class T>
struct RawAlloc {
    typedef T value_type;
    RawAlloc() = default;
    template <class U> RawAlloc(const RawAlloc<U>&) {}
    // big arrays: 2 MB-aligned and advised for transparent huge pages (a fresh 100+ MB array is otherwise
    // dominated by 4 KB page faults the first time it is written)
    static constexpr size_t HUGE_MIN = (size_t)8 << 20, HUGE_PAGE = (size_t)2 << 20;
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes < HUGE_MIN) return static_cast<T*>(::operator new(bytes));
        const size_t len = (bytes + HUGE_PAGE - 1) & ~(HUGE_PAGE - 1);
        void* p = std::aligned_alloc(HUGE_PAGE, len);
        if (!p) throw std::bad_alloc();
        (void)::madvise(p, len, MADV_HUGEPAGE);
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t n) {
        if (n * sizeof(T) < HUGE_MIN) ::operator delete(p); else std::free(p);
    }
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
        else ::new ((void*)p) U(std::forward<A>(a)...);
    }
    template <class U> bool operator==(const RawAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const RawAlloc<U>&) const { return false; }
};

// One constraint system in "dictionary order": for every row part the entries appear in the
// order the reference's DefaultDict would iterate them, explicit zero coefficients included
// (an empty part is the single entry {1 => 0}, ParseR1CS.jl:113-115).
struct Rows {
    std::vector<uint64_t, RawAlloc<uint64_t>> ptr[3];     // size nC+1 each, offsets into var/coef
    std::vector<uint32_t, RawAlloc<uint32_t>> var[3];     // 1-based variable id (= wire id + 1)
    std::vector<fp::u256, RawAlloc<fp::u256>> coef[3];    // canonical residue
    size_t n() const { return ptr[0].empty() ? 0 : ptr[0].size() - 1; }
    void start() {
        for (int p = 0; p < 3; ++p) { ptr[p].assign(1, 0); var[p].clear(); coef[p].clear(); }
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
    bool host_rows = false;      // `rows` is filled (false: the file was parsed on the device only; rows are fetched on demand)
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
// header + section table of a mapped file; cons_off = offset of the first constraint
inline int read_r1cs_header(const FileView& fv, R1CSFile& out, size_t& cons_off) {
    if (!fv.ok) return K_EIO;
    const size_t N = fv.size;
    const uint8_t* b = fv.data;
    auto need = [&](size_t off, size_t len) { return len <= N && off <= N - len; };   // (no wrap-around)
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
        const uint64_t sz = rd64(b + cur + 4);
        if (sz > ~(uint64_t)0 - cur - 12) return K_EFORMAT;      // a section size that would wrap the cursor around
        cur += 12 + (size_t)sz;
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
    cons_off = start[2];
    out.knowns.assign(1, 1);
    for (int64_t i = 2 + (int64_t)out.n_pub_out; i <= 1 + (int64_t)out.n_pub_out + out.n_pub_in + out.n_prv_in; ++i)
        out.knowns.push_back(i);
    out.outputs.clear();
    for (int64_t i = 2; i <= 1 + (int64_t)out.n_pub_out; ++i) out.outputs.push_back(i);
    out.n_vars = (int64_t)out.n_wires + 1;
    return K_OK;
}

inline int read_r1cs_rows(const FileView& fv, size_t cons_off, const char* path, R1CSFile& out);
inline int load_r1cs(const char* path, R1CSFile& out) {
    FileView fv(path);
    size_t cons_off = 0;
    const int st = read_r1cs_header(fv, out, cons_off);
    if (st != K_OK) return st;
    return read_r1cs_rows(fv, cons_off, path, out);
}
// the constraint section -> rows in dictionary order (host path)
inline int read_r1cs_rows(const FileView& fv, size_t cons_off, const char* path, R1CSFile& out) {
    const size_t N = fv.size;
    const uint8_t* b = fv.data;
    auto need = [&](size_t off, size_t len) { return len <= N && off <= N - len; };   // (no wrap-around)
    size_t start[4] = {0, 0, cons_off, 0};

    // pass 1 over the constraint section (sequential: record lengths are only known by walking them): term
    // counts per part, bounds check, and where every block of PARSE_BLOCK rows starts in the file and in the arrays
    const size_t PARSE_BLOCK = 2048;
    const size_t nblk = ((size_t)out.n_cons + PARSE_BLOCK - 1) / PARSE_BLOCK;
    std::vector<size_t> blk_off(nblk + 1);
    std::vector<uint64_t> blk_pos[3];
    for (int p = 0; p < 3; ++p) blk_pos[p].resize(nblk + 1);
    uint64_t terms[3] = {0, 0, 0};
    {
        size_t c = start[2];
        for (uint32_t r = 0; r < out.n_cons; ++r) {
            if (r % PARSE_BLOCK == 0) {
                blk_off[r / PARSE_BLOCK] = c;
                for (int p = 0; p < 3; ++p) blk_pos[p][r / PARSE_BLOCK] = terms[p];
            }
            for (int p = 0; p < 3; ++p) {
                if (!need(c, 4)) return K_EFORMAT;
                const uint32_t n = rd32(b + c);
                c += 4;
                if (!need(c, (size_t)n * 36)) return K_EFORMAT;
                c += (size_t)n * 36;
                terms[p] += n ? n : 1;   // an empty part is stored as {1 => 0}
            }
        }
        blk_off[nblk] = c;
        for (int p = 0; p < 3; ++p) blk_pos[p][nblk] = terms[p];
    }
    Rows& R = out.rows;
    for (int p = 0; p < 3; ++p) {
        R.ptr[p].resize((size_t)out.n_cons + 1);
        R.ptr[p][0] = 0;
        R.var[p].resize(terms[p]);
        R.coef[p].resize(terms[p]);
    }
    out.path = path;
    out.csr_built = false;
    // pass 2, one block of rows per task: every block writes at the positions pass 1 assigned to it. A part
    // that repeats a wire id comes out shorter than its term count ("last wins"); blocks where that
    // happened are closed up afterwards.
    std::vector<uint64_t> blk_len[3];
    for (int p = 0; p < 3; ++p) blk_len[p].resize(nblk);
    std::atomic<uint64_t> nnz_all[3];
    for (int p = 0; p < 3; ++p) nnz_all[p].store(0);
    for_chunks(nblk, [&](size_t blk, unsigned) {
        jl::SlotTable tab;
        std::vector<fp::u256> tmpc;
        size_t c = blk_off[blk];
        uint64_t pos[3] = {blk_pos[0][blk], blk_pos[1][blk], blk_pos[2][blk]}, nnz[3] = {0, 0, 0};
        const size_t r0 = blk * PARSE_BLOCK, r1 = std::min<size_t>(r0 + PARSE_BLOCK, out.n_cons);
        for (size_t r = r0; r < r1; ++r)
            for (int p = 0; p < 3; ++p) {
                const uint32_t n = rd32(b + c);
                c += 4;
                if (n == 0) {
                    R.var[p][pos[p]] = 1;
                    R.coef[p][pos[p]] = fp::make(0);
                    ++pos[p];
                } else if (n == 1) {   // one term: no dictionary order to reproduce
                    const fp::u256 v = fp::reduce(fp::make(rd64(b + c + 4), rd64(b + c + 12), rd64(b + c + 20), rd64(b + c + 28)));
                    R.var[p][pos[p]] = rd32(b + c) + 1;
                    R.coef[p][pos[p]] = v;
                    ++pos[p];
                    c += 36;
                    if (!fp::is_zero(v)) nnz[p]++;
                } else {
                    tab.reset();
                    tmpc.clear();
                    for (uint32_t k = 0; k < n; ++k) {
                        const uint32_t wire = rd32(b + c);
                        const fp::u256 v = fp::reduce(fp::make(rd64(b + c + 4), rd64(b + c + 12), rd64(b + c + 20), rd64(b + c + 28)));
                        c += 36;
                        bool ins;
                        int64_t& slot = tab.upsert((int64_t)wire + 1, (int64_t)tmpc.size(), ins);
                        if (ins) tmpc.push_back(v); else tmpc[(size_t)slot] = v;
                    }
                    tab.for_each([&](int64_t key, int64_t pay) {
                        R.var[p][pos[p]] = (uint32_t)key;
                        R.coef[p][pos[p]] = tmpc[(size_t)pay];
                        ++pos[p];
                        if (!fp::is_zero(tmpc[(size_t)pay])) nnz[p]++;
                    });
                }
                R.ptr[p][r + 1] = pos[p];
            }
        for (int p = 0; p < 3; ++p) {
            blk_len[p][blk] = pos[p] - blk_pos[p][blk];
            nnz_all[p].fetch_add(nnz[p], std::memory_order_relaxed);
        }
    });
    for (int p = 0; p < 3; ++p) {
        out.nnz[p] = nnz_all[p].load();
        uint64_t shift = 0;   // entries saved by repeated wire ids in the blocks before this one
        for (size_t blk = 0; blk < nblk; ++blk) {
            if (shift) {
                const uint64_t from = blk_pos[p][blk], len = blk_len[p][blk];
                std::memmove(R.var[p].data() + (from - shift), R.var[p].data() + from, len * sizeof(uint32_t));
                std::memmove(R.coef[p].data() + (from - shift), R.coef[p].data() + from, len * sizeof(fp::u256));
                const size_t r0 = blk * PARSE_BLOCK, r1 = std::min<size_t>(r0 + PARSE_BLOCK, out.n_cons);
                for (size_t r = r0; r < r1; ++r) R.ptr[p][r + 1] -= shift;
            }
            shift += (blk_pos[p][blk + 1] - blk_pos[p][blk]) - blk_len[p][blk];
        }
        if (shift) {
            R.var[p].resize(terms[p] - shift);
            R.coef[p].resize(terms[p] - shift);
        }
    }
    out.host_rows = true;
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
    auto need = [&](size_t off, size_t len) { return len <= N && off <= N - len; };   // (no wrap-around)
    if (!need(0, 12)) return K_EFORMAT;
    size_t cur = 12, start2 = 0;
    for (int s = 0; s < 3; ++s) {
        if (!need(cur, 12)) return K_EFORMAT;
        if (rd32(b + cur) == 2) start2 = cur + 12;
        const uint64_t sz = rd64(b + cur + 4);
        if (sz > ~(uint64_t)0 - cur - 12) return K_EFORMAT;
        cur += 12 + (size_t)sz;
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
// Appearance signatures of the variables of one window: for every variable, in first-seen (dictionary) order,
// the list of (part counter, coefficient) it occurs with. Flat storage: occurrences are appended as they come
// and grouped per variable by finish() (a counting sort, so each list keeps its order) -- a window of 16 k
// rows costs a handful of array growths instead of one small vector per variable.
struct AppearMap {
    jl::SlotTable tab;                       // variable -> list number
    std::vector<uint32_t> occ_list;          // per occurrence, in arrival order
    std::vector<int64_t> occ_where;
    std::vector<const fp::u256*> occ_coef;
    std::vector<uint32_t> start, slot;       // after finish(): list i = occurrences slot[start[i] .. start[i+1])
    size_t used = 0;
    void clear() {
        tab.reset();
        used = 0;
        occ_list.clear(); occ_where.clear(); occ_coef.clear();
    }
    void add(int64_t var, int64_t where, const fp::u256* c) {
        bool ins;
        int64_t& s = tab.upsert(var, (int64_t)used, ins);
        if (ins) ++used;
        occ_list.push_back((uint32_t)s);
        occ_where.push_back(where);
        occ_coef.push_back(c);
    }
    void finish() {
        start.assign(used + 1, 0);
        for (uint32_t l : occ_list) start[l + 1]++;
        for (size_t i = 0; i < used; ++i) start[i + 1] += start[i];
        slot.resize(occ_list.size());
        std::vector<uint32_t> fill(start.begin(), start.end() - 1);
        for (size_t o = 0; o < occ_list.size(); ++o) slot[fill[occ_list[o]]++] = (uint32_t)o;
    }
    size_t len(size_t l) const { return start[l + 1] - start[l]; }
    int64_t where(size_t l, size_t i) const { return occ_where[slot[start[l] + i]]; }
    const fp::u256& coef(size_t l, size_t i) const { return *occ_coef[slot[start[l] + i]]; }
    // (variable, list number) sorted by list, stable w.r.t. table order
    std::vector<std::pair<int64_t, int64_t>> sorted() const {
        std::vector<std::pair<int64_t, int64_t>> v;
        v.reserve(used);
        tab.for_each([&](int64_t key, int64_t pay) { v.push_back({key, pay}); });
        std::stable_sort(v.begin(), v.end(), [&](const std::pair<int64_t, int64_t>& a, const std::pair<int64_t, int64_t>& b) {
            return compare(*this, (size_t)a.second, *this, (size_t)b.second) < 0;
        });
        return v;
    }
    // lexicographic order of two lists: by (counter, coefficient) pairs, a proper prefix first
    static int compare(const AppearMap& X, size_t x, const AppearMap& Y, size_t y) {
        const size_t nx = X.len(x), ny = Y.len(y), n = std::min(nx, ny);
        for (size_t i = 0; i < n; ++i) {
            const int64_t wx = X.where(x, i), wy = Y.where(y, i);
            if (wx != wy) return wx < wy ? -1 : 1;
            const int c = fp::cmp(X.coef(x, i), Y.coef(y, i));
            if (c) return c;
        }
        return nx < ny ? -1 : nx > ny ? 1 : 0;
    }
};
}  // namespace detail

// ---- abstraction in three steps (so that the device front-end, frontend.hpp, can replace the first two and keep the third)
// The trusted function's side of the comparison, prepared once per abstraction: its appearance signatures sorted (l2), its
// per-part sorted coefficient lists, and where each of its variables sits in l2.
struct PatternHost {
    detail::AppearMap orig;
    std::vector<std::pair<int64_t, int64_t>> l2;
    std::vector<fp::u256> subvals;
    std::vector<size_t> subptr;
    std::unordered_map<int64_t, size_t> where;   // pattern variable -> position in l2 (later duplicates win, as in the map it replaces)
    size_t nS = 0;
    void build(const R1CSFile& sub) {
        using namespace detail;
        nS = sub.rows.n();
        int64_t counter = 1;
        for (size_t j = 0; j < nS; ++j)
            for (int p = 0; p < 3; ++p) {
                for (uint64_t k = sub.rows.ptr[p][j]; k < sub.rows.ptr[p][j + 1]; ++k)
                    if (!fp::is_zero(sub.rows.coef[p][k])) orig.add(sub.rows.var[p][k], counter, &sub.rows.coef[p][k]);
                ++counter;
            }
        orig.finish();
        l2 = orig.sorted();
        subptr.assign(1, 0);
        std::vector<fp::u256> vb;
        for (size_t j = 0; j < nS; ++j)
            for (int p = 0; p < 3; ++p) {
                part_values(sub.rows, p, j, vb);
                subvals.insert(subvals.end(), vb.begin(), vb.end());
                subptr.push_back(subvals.size());
            }
        for (size_t x = 0; x < l2.size(); ++x) where[l2[x].first] = x;
    }
};
// Is the window rows[at, at + nS) an occurrence of the pattern (:272-351)? On success image[x] = the window's variable for the
// pattern variable l2[x]. cur / va: scratch of the calling worker.
inline bool verify_window(const PatternHost& P, const Rows& rows, size_t at, detail::AppearMap& cur, std::vector<fp::u256>& va,
                          std::vector<int64_t>& image) {
    using namespace detail;
    cur.clear();
    int64_t counter = 0;
    for (size_t j = 0; j < P.nS; ++j)
        for (int p = 0; p < 3; ++p) {
            ++counter;
            part_values(rows, p, at + j, va);
            const size_t s0 = P.subptr[j * 3 + (size_t)p], s1 = P.subptr[j * 3 + (size_t)p + 1];
            if (va.size() != s1 - s0) return false;
            for (size_t t = 0; t < va.size(); ++t)
                if (!fp::eq(va[t], P.subvals[s0 + t])) return false;
            for (uint64_t k = rows.ptr[p][at + j]; k < rows.ptr[p][at + j + 1]; ++k)
                if (!fp::is_zero(rows.coef[p][k])) cur.add(rows.var[p][k], counter, &rows.coef[p][k]);
        }
    cur.finish();
    const auto l1 = cur.sorted();
    if (l1.size() != P.l2.size()) return false;
    for (size_t x = 0; x < l1.size(); ++x)
        if (AppearMap::compare(cur, (size_t)l1[x].second, P.orig, (size_t)P.l2[x].second) != 0) return false;
    image.resize(l1.size());
    for (size_t x = 0; x < l1.size(); ++x) image[x] = l1[x].first;
    return true;
}
// Greedy left to right with the reference's stuck cursor (:368-388): a match that starts inside the previous window is never
// reached again, and neither is any later one. Rows outside the replaced windows survive as ranges [a, b) in `keep`; the new
// special constraints go to `fresh`. image_of(ci, pattern variable, &window variable) -> false = KeyError (:381-382).
template <class ImageOf>
inline int greedy_replace(const std::string& name, const R1CSFile& sub, size_t nC, const std::vector<size_t>& cand,
                          const std::vector<uint8_t>& matched, ImageOf&& image_of, std::vector<std::pair<size_t, size_t>>& keep,
                          std::vector<Special>& fresh) {
    const size_t nS = sub.rows.n();
    size_t i = 0;
    for (size_t ci = 0; ci <= cand.size(); ++ci) {
        if (ci < cand.size() && !matched[ci]) continue;
        const size_t stop = ci < cand.size() ? cand[ci] : nC;
        if (stop < i) { keep.push_back({i, nC}); i = nC; break; }
        keep.push_back({i, stop});
        i = stop;
        if (ci == cand.size()) break;
        Special sp;
        sp.name = name;
        for (int64_t x : sub.knowns)
            if (x != 1) {
                int64_t v;
                if (!image_of(ci, x, v)) return K_EKEY;
                sp.inputs.push_back(v);
            }
        for (int64_t x : sub.outputs) {
            int64_t v;
            if (!image_of(ci, x, v)) return K_EKEY;
            sp.outputs.push_back(v);
        }
        fresh.push_back(std::move(sp));
        i += nS;
    }
    return K_OK;
}

// Replaces every (greedy, left to right, stuck-cursor) occurrence of `sub` in `rows` by a special
// constraint: the reduced rows go to `red`, the new specials are appended. Returns K_OK or K_EKEY (then
// `red` and `specials` are left as they were).  Fingerprints, the per-window variable matching and the
// copy of the surviving rows run on the host worker threads; which windows match, and in which order,
// does not depend on the thread count.
// device_cand: candidate window starts found on the GPU (abstract.hip.hpp), ascending -- a superset of the occurrences, like
// the host scan's; null = scan here.
inline int abstract_one(const std::string& name, const Rows& rows, const R1CSFile& sub, std::vector<Special>& specials, Rows& red,
                        const std::vector<size_t>* device_cand = nullptr) {
    using namespace detail;
    const size_t nC = rows.n(), nS = sub.rows.n();
    const size_t FP_BLOCK = 8192;
    std::vector<size_t> cand;
    if (device_cand) cand = *device_cand;
    else if (nC >= nS) {
        std::vector<uint64_t> fb(nC), fs(nS);
        for_chunks((nC + FP_BLOCK - 1) / FP_BLOCK, [&](size_t blk, unsigned) {
            std::vector<fp::u256> scratch;
            const size_t i1 = std::min(nC, (blk + 1) * FP_BLOCK);
            for (size_t i = blk * FP_BLOCK; i < i1; ++i) fb[i] = row_fingerprint(rows, i, scratch);
        });
        std::vector<fp::u256> scratch;
        for (size_t i = 0; i < nS; ++i) fs[i] = row_fingerprint(sub.rows, i, scratch);
        for (size_t i = 0; i + nS <= nC; ++i) {
            bool m = true;
            for (size_t j = 0; j + 1 < nS; ++j)
                if (fb[i + j] != fs[j]) { m = false; break; }
            if (m) cand.push_back(i);
        }
    }
    PatternHost P;
    P.build(sub);
    // one candidate window per task; image[c] = the window's variable for every pattern variable in l2 order
    std::vector<std::vector<int64_t>> image(cand.size());
    std::vector<uint8_t> matched(cand.size(), 0);
    {
        const unsigned W = for_chunks_workers(cand.size());
        std::vector<AppearMap> curs(W);
        std::vector<std::vector<fp::u256>> vas(W);
        for_chunks(cand.size(), [&](size_t ci, unsigned w) { matched[ci] = verify_window(P, rows, cand[ci], curs[w], vas[w], image[ci]) ? 1 : 0; });
    }
    std::vector<std::pair<size_t, size_t>> keep;
    std::vector<Special> fresh;
    const int rc = greedy_replace(name, sub, nC, cand, matched, [&](size_t ci, int64_t x, int64_t& v) {
        auto it = P.where.find(x);
        if (it == P.where.end()) return false;
        v = image[ci][it->second];
        return true;
    }, keep, fresh);
    if (rc != K_OK) return rc;
    // lay the surviving ranges out back to back, then copy them in pieces on the worker threads
    struct Piece { size_t a, b, row; uint64_t at[3]; };
    const size_t COPY_BLOCK = 16384;
    std::vector<Piece> pieces;
    size_t nrow = 0;
    uint64_t nterm[3] = {0, 0, 0};
    for (auto& kb : keep)
        for (size_t a = kb.first; a < kb.second; a += COPY_BLOCK) {
            Piece pc;
            pc.a = a;
            pc.b = std::min(kb.second, a + COPY_BLOCK);
            pc.row = nrow;
            for (int p = 0; p < 3; ++p) {
                pc.at[p] = nterm[p];
                nterm[p] += rows.ptr[p][pc.b] - rows.ptr[p][pc.a];
            }
            nrow += pc.b - pc.a;
            pieces.push_back(pc);
        }
    for (int p = 0; p < 3; ++p) {
        red.ptr[p].resize(nrow + 1);
        red.ptr[p][0] = 0;
        red.var[p].resize(nterm[p]);
        red.coef[p].resize(nterm[p]);
    }
    for_chunks(pieces.size(), [&](size_t pi, unsigned) {
        const Piece& pc = pieces[pi];
        for (int p = 0; p < 3; ++p) {
            const uint64_t k0 = rows.ptr[p][pc.a], k1 = rows.ptr[p][pc.b];
            if (k1 > k0) {
                std::memcpy(red.var[p].data() + pc.at[p], rows.var[p].data() + k0, (k1 - k0) * sizeof(uint32_t));
                std::memcpy(red.coef[p].data() + pc.at[p], rows.coef[p].data() + k0, (k1 - k0) * sizeof(fp::u256));
            }
            for (size_t r = pc.a; r < pc.b; ++r) red.ptr[p][pc.row + (r - pc.a) + 1] = rows.ptr[p][r + 1] - k0 + pc.at[p];
        }
    });
    for (auto& sp : fresh) specials.push_back(std::move(sp));
    return K_OK;
}

}  // namespace ecne
