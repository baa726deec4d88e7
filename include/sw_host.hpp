// sw_host.hpp -- C++ mirror of the reference's function objects for the hot path, header-only over the C ABI
// (include/swmi.h).  Same names, argument meaning and return shapes as the Java classes, so C++ host code reads
// like the reference:
//
//   sw::SmithWaterman::OptAlignments().call({ref, read}, {5,-3,-4}, {'a','i','d','-'})   src/sw/SmithWaterman.java:35,62-92
//   sw::DistributedSW::OptAlignments  (strict '>' tie order)                              src/sw/DistributedSW.java:50,77-104
//   sw::Distribution::MapRef().call({ref{metadata,sequence}, reads, algoArgs})             src/sw/Distribution.java:383,403-436
//   sw::Distribution::MapPartition().call(tuples)   one native call for a whole partition
//   sw::Distribution::CombineReadsToRef().call(refs, reads, algoArgs)                     src/sw/Distribution.java:702-725
//
// Java's Tuple2<Integer, ArrayList<Tuple2<Integer,String[]>>> becomes std::pair<int, std::vector<MatchSite>>.
// All alignment work happens in libswmi.so on the GPU; errors surface as std::runtime_error(swmi_last_error()).
#pragma once
#include <array>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "swmi.h"

namespace sw {

using MatchSite = std::pair<int, std::array<std::string, 2>>;       // Tuple2<Integer, String[]{refAligned, readAligned}>
using AlgoArgs = std::pair<std::array<int, 3>, std::array<char, 4>>;  // Tuple2<int[]{match,mismatch,gap}, char[]{a,i,d,-}>

inline const std::array<int, 3> ALIGN_SCORES{5, -3, -4};             // Distribution.java:36
inline const std::array<char, 4> ALIGN_TYPES{'a', 'i', 'd', '-'};    // Distribution.java:37

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
inline void check(int rc) { if (rc != SWMI_OK) throw Error(swmi_last_error()); }

class Context {                    // one GPU + one HIP stream; use one per host thread
public:
    explicit Context(int device = 0) { check(swmi_create(device, &ctx_)); }
    ~Context() { swmi_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    swmi_ctx *get() const { return ctx_; }
private:
    swmi_ctx *ctx_ = nullptr;
};

namespace detail {
inline swmi_params params(const std::array<int, 3> &sc, const std::array<char, 4> &ty, int tie) {
    swmi_params p;
    swmi_default_params(&p);
    p.match = sc[0]; p.mismatch = sc[1]; p.gap = sc[2]; p.tie_mode = tie;
    for (int k = 0; k < 4; k++) p.types[k] = ty[k];
    return p;
}
struct Packed { std::string blob; std::vector<uint64_t> off{0}; };
inline Packed pack(const std::vector<std::string> &v) {
    Packed p;
    for (auto &s : v) { p.blob += s; p.off.push_back(p.blob.size()); }
    return p;
}
struct Batch {                     // RAII over swmi_batch
    swmi_ctx *ctx; swmi_batch *b = nullptr;
    Batch(swmi_ctx *c, const std::vector<std::string> &refs, const std::vector<std::string> &reads, const swmi_params &p) : ctx(c) {
        Packed r = pack(refs), q = pack(reads);
        check(swmi_align_batch(c, &p, (const uint8_t *)r.blob.data(), r.off.data(), (uint32_t)refs.size(),
                               (const uint8_t *)q.blob.data(), q.off.data(), (uint32_t)reads.size(), &b));
    }
    ~Batch() { swmi_batch_free(ctx, b); }
};
}  // namespace detail

struct SmithWaterman {
    struct OptAlignments {
        int tie_mode = SWMI_TIE_SERIAL;
        Context &ctx;
        explicit OptAlignments(Context &c) : ctx(c) {}
        // seqs = {reference, read}
        std::pair<int, std::vector<MatchSite>> call(const std::array<std::string, 2> &seqs,
                                                    const std::array<int, 3> &alignScores = ALIGN_SCORES,
                                                    const std::array<char, 4> &alignTypes = ALIGN_TYPES) const {
            detail::Batch b(ctx.get(), {seqs[0]}, {seqs[1]}, detail::params(alignScores, alignTypes, tie_mode));
            int32_t score = 0; uint64_t n = 0; uint32_t flags = 0;
            check(swmi_pair_score(b.b, 0, &score));
            check(swmi_pair_n_alignments(b.b, 0, &n, &flags));
            std::vector<MatchSite> out(n);
            for (uint64_t k = 0; k < n; k++) {
                int32_t begin; const char *r, *q;
                check(swmi_pair_alignment(b.b, 0, k, &begin, nullptr, nullptr, &r, &q, nullptr));
                out[k] = {begin, {r, q}};
            }
            return {score, std::move(out)};
        }
    };
};

struct DistributedSW {
    struct OptAlignments : SmithWaterman::OptAlignments {
        explicit OptAlignments(Context &c) : SmithWaterman::OptAlignments(c) { tie_mode = SWMI_TIE_STRICT; }
    };
};

struct Distribution {
    using Ref = std::array<std::string, 2>;                                   // {metadata, sequence}
    struct Tuple3 { Ref ref; const std::vector<std::string> *reads; AlgoArgs algoArgs; };
    using MapResult = std::pair<int, std::pair<Ref, std::vector<MatchSite>>>;   // (total, (ref, matchSites))

    struct CombineReadsToRef {
        std::vector<Tuple3> call(const std::vector<Ref> &references, const std::vector<std::string> &reads,
                                 const AlgoArgs &algoArgs = {ALIGN_SCORES, ALIGN_TYPES}) const {
            std::vector<Tuple3> v;
            for (auto &r : references) v.push_back({r, &reads, algoArgs});
            return v;
        }
    };

    struct MapPartition {          // every tuple of a partition (same reads and algoArgs) in ONE native call
        Context &ctx;
        int tie_mode = SWMI_TIE_SERIAL;
        explicit MapPartition(Context &c) : ctx(c) {}
        std::vector<MapResult> call(const std::vector<Tuple3> &tuples) const {
            std::vector<MapResult> out;
            if (tuples.empty()) return out;
            std::vector<std::string> refs;
            for (auto &t : tuples) refs.push_back(t.ref[1]);
            detail::Batch b(ctx.get(), refs, *tuples[0].reads,
                            detail::params(tuples[0].algoArgs.first, tuples[0].algoArgs.second, tie_mode));
            for (uint32_t r = 0; r < tuples.size(); r++) {
                int32_t total; uint64_t n;
                check(swmi_ref_total(b.b, r, &total));                          // Distribution.java:424
                check(swmi_ref_n_match_sites(b.b, r, &n));                      // :425-428
                std::vector<MatchSite> sites(n);
                for (uint64_t k = 0; k < n; k++) {
                    int32_t begin; const char *ra, *qa;
                    check(swmi_ref_match_site(b.b, r, k, &begin, &ra, &qa, nullptr));
                    sites[k] = {begin, {ra, qa}};
                }
                out.push_back({total, {tuples[r].ref, std::move(sites)}});
            }
            return out;
        }
    };

    struct MapRef {                // per-element form, same shape as the reference's PairFunction
        Context &ctx;
        explicit MapRef(Context &c) : ctx(c) {}
        MapResult call(const Tuple3 &t) const { return MapPartition(ctx).call({t})[0]; }
    };
};

}  // namespace sw
