// frontend.hpp — interface between the two translation units of libecne_hip: ecne_engine.hip (C ABI, solve) and
// ecne_frontend.hip (the device front-end: .r1cs parse, abstraction and flat-array layout on the GPU).
//
// What the front-end replaces, by behaviour:
//   readR1CS                 /root/reference/src/ParseR1CS.jl:50-124      -> fe::parse_on_device
//   abstraction              /root/reference/src/R1CSConstraintSolver.jl:237-395 -> fe::abstract_on_device
//   the per-solve set-up of SolveConstraintsSymbolic that only depends on the rows (nzk_a/b/c :698-700,
//   variable_to_indices :628-633, the shape tests the rules repeat on every visit)   -> fe::layout_on_device
// The host implementations of the same three steps (host_model.hpp, build_layout in ecne_engine.hip) stay: they serve small
// files, the lazily built host views (ecne_system_rows, the report orders) and every case the device path hands back
// (FE_FALLBACK); tests/test_gpu_frontend.py compares the two array by array.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "engine_types.hpp"
#include "host_model.hpp"

namespace ecne {
namespace fe {

enum : int { FE_FALLBACK = 1 };   // not an error: "this input is for the host path" (returned instead of a Status)

// Rows in dictionary order resident on one device: the device twin of host_model.hpp's Rows (same arrays, same order).
struct DevRows {
    int device = -1;
    uint64_t n = 0;
    void* arena = nullptr;
    size_t arena_bytes = 0;
    uint64_t* ptr[3] = {nullptr, nullptr, nullptr};     // n + 1 offsets per part
    uint32_t* var[3] = {nullptr, nullptr, nullptr};
    uint64_t* coef[3] = {nullptr, nullptr, nullptr};    // 4 limbs per entry, canonical
    uint64_t terms[3] = {0, 0, 0};                      // entries per part (explicit zeros and {1 => 0} placeholders included)
    uint64_t nnz[3] = {0, 0, 0};                        // non-zero coefficients per part (parse only; not kept by abstraction)
    ~DevRows();
    DevRows() = default;
    DevRows(const DevRows&) = delete;
    DevRows& operator=(const DevRows&) = delete;
};

struct ParseStats {
    double upload_ms = 0, offsets_ms = 0, fill_ms = 0, total_ms = 0;
    uint64_t file_bytes = 0;
};
// `file` = the mapped file, `cons_off` = offset of the first constraint (the reader walks n_cons rows from the section start and is
// bounded by the end of the FILE, not by the section size). K_OK, K_EFORMAT (the walk leaves the file), K_ENODEVICE, or FE_FALLBACK
// (input the device path does not take: 4 GiB and more, parts of 2^18 terms and more, a hash table that grows past its scratch).
int parse_on_device(const uint8_t* file, size_t file_size, size_t cons_off, uint32_t n_cons, int device, std::shared_ptr<DevRows>& out, ParseStats& st);
int download_rows(const DevRows& D, Rows& out);
int warmup(int device);      // loads the front-end's code object (ecne_warmup)

struct AbstractDevStats {
    double fp_ms = 0, scan_ms = 0, verify_ms = 0, compact_ms = 0, prep_ms = 0;
    uint64_t bytes = 0;
    size_t n_cand = 0, n_matched = 0, n_host_verified = 0;
};
// abstraction() with everything that scales with the main file on the device: fingerprints + window scan (abstract.hip.hpp),
// exact verification of the candidate windows (variable bijection found by signature hashes, PROVEN entry by entry), greedy
// replacement decided on the host from the per-window verdicts, surviving rows compacted on the device. `sub` needs host rows.
// Returns K_OK (specials appended, `red` = reduced rows, possibly `rows` itself), K_EKEY, K_ENODEVICE, or FE_FALLBACK.
int abstract_on_device(const std::string& name, const std::shared_ptr<DevRows>& rows, const R1CSFile& sub, std::vector<Special>& specials,
                       std::shared_ptr<DevRows>& red, AbstractDevStats& st);

struct LayoutCounts {
    uint32_t nC = 0, nVall = 0;      // rows; largest variable id the state arrays must hold
    uint64_t nnz[3] = {0, 0, 0};
    uint32_t n_vals = 0, nP4 = 0, nP5 = 0, nCls = 0, nLong = 0, nBigRows = 0, fo_total = 0, maxrowC = 0;
    uint32_t dsu_err = 0;            // some row makes secp_solve's dsu setup raise (:650-656)
};
// what the layout kernels write: the static arrays of a system that only depend on its rows (device pointers)
struct LayoutDst {
    uint32_t* rp[3]; uint32_t* col[3]; uint64_t* coef[3];
    RowInfo* rinfo;
    RowInfo* rinfo0;              // the structural descriptors as laid down, before k_classify_rows completes them (ecne_classify restores from it)
    uint32_t *fo_ptr, *fo_rows;
    uint8_t* nontrivial;          // every variable of a non-zero term; the caller adds specials' and target ids (mark_bytes)
    uint32_t *p4_list, *p4_b, *p4_s, *cls_list;
    uint16_t* tbig;
    uint32_t *bigrows, *long_list, *p5_rows, *p5_y, *rec, *foi;
};
struct LayoutDev {                // owns the device memory LayoutDst points into
    int device = -1;
    LayoutCounts cnt;
    LayoutDst dst;
    void* mem[3] = {nullptr, nullptr, nullptr};
    double ms = 0;
    ~LayoutDev();
    LayoutDev() = default;
    LayoutDev(const LayoutDev&) = delete;
    LayoutDev& operator=(const LayoutDev&) = delete;
};
// n_vars: nWires + 1; min_nv: largest variable id the specials mention (state arrays are sized for the largest id seen).
// K_OK, K_ENODEVICE / K_ECAPACITY, or FE_FALLBACK (sizes the device path does not take).
int layout_on_device(const DevRows& D, uint32_t n_vars, uint32_t min_nv, std::unique_ptr<LayoutDev>& out);
// nontrivial[id] = 1 for the listed ids (specials' inputs / outputs, targets: :600-618)
int mark_bytes(int device, uint8_t* dst, const std::vector<uint32_t>& ids);
// rows that still hold a variable without the `unique` bit ("Bad Constraints", :1609-1618), 1-based, ascending
int bad_rows(int device, const Job& J, std::vector<int64_t>& out);
// abstraction's candidate scan for a system whose rows live on the host (coefficients uploaded first)
int candidates_for_host_rows(const Rows& rows, const Rows& sub, int device, std::vector<size_t>& cand, AbstractDevStats& st, double& upload_ms);

// ---- signature hash shared by the host-side pattern preparation and the device verification kernels
#if defined(__HIPCC__)
#define FEQ __host__ __device__ __forceinline__
#else
#define FEQ inline
#endif
FEQ uint64_t sig_mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
// one (part counter, coefficient) appearance of a variable (:334-351); a variable's signature hash is the SUM of these
FEQ void sig_hash(uint64_t counter, const uint64_t* c, uint64_t& h1, uint64_t& h2) {
    uint64_t a = sig_mix(counter * 0x9e3779b97f4a7c15ULL + 0x1234567ULL);
    uint64_t b = sig_mix(counter ^ 0xc2b2ae3d27d4eb4fULL);
    for (int i = 0; i < 4; ++i) {
        a = sig_mix(a ^ c[i]);
        b = sig_mix(b + c[i] * 0x165667b19e3779f9ULL + (uint64_t)i);
    }
    h1 = a;
    h2 = b | 1ull;
}

}  // namespace fe
}  // namespace ecne
