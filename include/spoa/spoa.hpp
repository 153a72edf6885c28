// spoa/spoa.hpp — product header of libgbx: the subset of the spoa v3 C++ API that GenomicsBench's poa driver
// uses (R/benchmarks/poa/msa_spoa_omp.cpp:20-22,37,189-194,237-252), implemented on top of the C-ABI of
// include/gbx.h so that the UNMODIFIED driver compiles against it and runs its windows on the MI355X.
//
//   spoa::createAlignmentEngine(type, m, n, g, e, q, c)   :189-190   throws std::invalid_argument (caught at :191)
//   spoa::createGraph()                                    :237
//   AlignmentEngine::align(sequence, graph)                :242      returns a token, see below
//   Graph::add_alignment(alignment, sequence)              :247
//   Graph::generate_consensus()                            :252      one gbx_poa_consensus_host call for the window
//
// The device boundary is the whole window: aligning one sequence at a time against a graph that
// lives on the host would ship the graph across PCIe once per sequence.  So align() does no arithmetic: it hands
// back a token that names the graph and the sequence's position in it, add_alignment() checks the token and records
// the sequence, and generate_consensus() sends the recorded window through gbx_poa_consensus_host, which performs
// align / add_alignment / generate_consensus for every sequence in order on the GPU.  The consensus string is the
// one spoa returns for the same calls.  What the token cannot carry is a caller that inspects or edits the
// alignment between align() and add_alignment(): add_alignment() throws std::invalid_argument for anything that is
// not the token of that graph's next sequence.  Not a spoa source file: written from the call sites above.
#ifndef GBX_SPOA_FACADE_HPP
#define GBX_SPOA_FACADE_HPP

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../gbx.h"

namespace spoa {

enum class AlignmentType { kSW = 0, kNW = 1, kOV = 2 };   // the driver passes static_cast<AlignmentType>(1), :184,190

using Alignment = std::vector<std::pair<std::int32_t, std::int32_t>>;

class Graph;
class AlignmentEngine;

std::unique_ptr<Graph> createGraph();
std::unique_ptr<AlignmentEngine> createAlignmentEngine(AlignmentType type, std::int8_t m, std::int8_t n, std::int8_t g,
                                                       std::int8_t e, std::int8_t q, std::int8_t c);

class Graph {
public:
    // Records `sequence` as the next one of the window.  `alignment` must be the token align() returned for this
    // sequence and this graph.
    void add_alignment(const Alignment &alignment, const std::string &sequence, std::uint32_t weight = 1)
    {
        if (weight != 1) throw std::invalid_argument("[spoa(gbx)::Graph::add_alignment] error: only unit weights (the driver's) are supported");
        if (alignment.size() != 1 || alignment[0].first != kToken || alignment[0].second != (std::int32_t)lens_.size())
            throw std::invalid_argument("[spoa(gbx)::Graph::add_alignment] error: the alignment is not the one align() returned for "
                                        "this graph's next sequence");
        if (sequence.empty()) return;                       // spoa ignores empty sequences
        offs_.push_back((std::int64_t)arena_.size());
        lens_.push_back((std::int32_t)sequence.size());
        arena_ += sequence;
        consensus_valid_ = false;
    }

    // align + add_alignment + consensus of the recorded sequences, in order, on the GPU.
    std::string generate_consensus()
    {
        if (lens_.empty()) return std::string();
        if (consensus_valid_) return consensus_;
        std::int32_t lmax = 0;
        for (std::int32_t l : lens_) if (l > lmax) lmax = l;
        const std::int64_t stride = 2 * (std::int64_t)lmax + 64;
        std::vector<char> cons((std::size_t)stride);
        std::int32_t cons_len = 0;
        const std::int64_t wf[2] = {0, (std::int64_t)lens_.size()};
        std::string padded = arena_ + std::string(16, '\0');
        const int rc = gbx_poa_consensus_host(&params_, 1, wf, (std::int64_t)lens_.size(), offs_.data(), lens_.data(),
                                              padded.data(), (std::int64_t)arena_.size(), cons.data(), &cons_len, stride);
        if (rc != GBX_OK) throw std::runtime_error(std::string("[spoa(gbx)::Graph::generate_consensus] error: ") + gbx_last_error());
        consensus_.assign(cons.data(), (std::size_t)cons_len);
        consensus_valid_ = true;
        return consensus_;
    }

    std::uint32_t num_sequences() const { return (std::uint32_t)lens_.size(); }

private:
    friend class AlignmentEngine;
    friend std::unique_ptr<Graph> createGraph();
    Graph() { gbx_poa_default_params(&params_); }
    static constexpr std::int32_t kToken = INT32_MIN;
    gbx_poa_params params_;
    bool have_params_ = false;
    std::string arena_;
    std::vector<std::int64_t> offs_;
    std::vector<std::int32_t> lens_;
    std::string consensus_;
    bool consensus_valid_ = false;
};

class AlignmentEngine {
public:
    // Returns the token add_alignment() expects; tells the graph which scores its window is aligned with.
    Alignment align(const std::string &sequence, const std::unique_ptr<Graph> &graph)
    {
        (void)sequence;
        if (!graph) throw std::invalid_argument("[spoa(gbx)::AlignmentEngine::align] error: null graph");
        if (graph->have_params_ && (graph->params_.m != params_.m || graph->params_.n != params_.n || graph->params_.g != params_.g ||
                                    graph->params_.e != params_.e || graph->params_.q != params_.q || graph->params_.c != params_.c))
            throw std::invalid_argument("[spoa(gbx)::AlignmentEngine::align] error: one graph, two engines with different scores");
        graph->params_ = params_;
        graph->have_params_ = true;
        return Alignment(1, std::make_pair(Graph::kToken, (std::int32_t)graph->lens_.size()));
    }
    void prealloc(std::uint32_t, std::uint32_t) {}          // spoa's optional reservation: nothing to reserve on the host

private:
    friend std::unique_ptr<AlignmentEngine> createAlignmentEngine(AlignmentType, std::int8_t, std::int8_t, std::int8_t,
                                                                  std::int8_t, std::int8_t, std::int8_t);
    explicit AlignmentEngine(const gbx_poa_params &p) : params_(p) {}
    gbx_poa_params params_;
};

inline std::unique_ptr<Graph> createGraph() { return std::unique_ptr<Graph>(new Graph()); }

inline std::unique_ptr<AlignmentEngine> createAlignmentEngine(AlignmentType type, std::int8_t m, std::int8_t n, std::int8_t g,
                                                              std::int8_t e, std::int8_t q, std::int8_t c)
{
    if (type != AlignmentType::kSW && type != AlignmentType::kNW && type != AlignmentType::kOV)
        throw std::invalid_argument("[spoa::createAlignmentEngine] error: invalid alignment type!");
    if (g > 0 || q > 0) throw std::invalid_argument("[spoa::createAlignmentEngine] error: gap opening penalty must be non-positive!");
    if (e > 0 || c > 0) throw std::invalid_argument("[spoa::createAlignmentEngine] error: gap extension penalty must be non-positive!");
    if (type != AlignmentType::kNW)
        throw std::invalid_argument("[spoa(gbx)::createAlignmentEngine] error: only global alignment (kNW, the driver's algorithm 1) runs on the device");
    gbx_poa_params p;
    gbx_poa_default_params(&p);
    p.m = m; p.n = n; p.g = g; p.e = e; p.q = q; p.c = c;
    return std::unique_ptr<AlignmentEngine>(new AlignmentEngine(p));
}

}  // namespace spoa
#endif
