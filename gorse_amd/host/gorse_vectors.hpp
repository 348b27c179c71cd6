// gorse_vectors.hpp -- C++ twin of the reference's vectors.Database interface (storage/vectors/database.go:107-120)
// with a backend for the "hip://" prefix: collections of DENSE vectors whose similarity search is the exact GPU top-k
// of libgorse_hip (gorse_topk_*), i.e. what storage/vectors/hip.go would contain behind `//go:build cgo && hip` and
// `vectors.Register([]string{"hip://"}, creator)` (database.go:155-175).
//
// Semantics follow the reference's own backend for dense collections (storage/vectors/xvec.go):
//   * AddVectors upserts by Id (xvec.go:301-327); GetVectors returns the found vectors in the order of the requested
//     ids, duplicates and misses dropped (database.go:136-153 orderVectors); DeleteVectors removes everything with
//     Timestamp < cutoff at millisecond resolution (xvec.go:371-377).
//   * QueryVectors (xvec.go:379-446): hidden vectors never match; `categories` is CONTAIN_ALL; topK <= 0 -> empty;
//     Score is "higher = more similar": the inner product for Dot, the NEGATED distance for Euclidean and Cosine
//     (xvec.go:425-427).  The query vector itself is not excluded (database_test.go:146-153).
//   * Search is EXACT here (the reference's DiskANN index is approximate: rank order on the reference's test inputs is
//     the pinned behaviour, numeric scores are not -- SURVEY.md 8c).  Distances are the reference's own functions:
//     -floats.Dot, floats.Euclidean, 1 - cos (gorse_hip.h GORSE_METRIC_*).
//   * Filters are applied after an over-fetched exact search: k' = topK + slack vectors are fetched, filtered, and k'
//     grows until topK admissible vectors are found or the collection is exhausted, so the answer is the exact top-K of
//     the admissible set.
//   * Sparse collections (dimension 0, `Indices`; distance Dot only, xvec.go:241-247) are searched by the exact sparse
//     top-k of libgorse_hip (gorse_sparse_*): every admissible vector ranked by its inner product, cut to topK, the
//     zero scores dropped (xvec.go:419-421).  The hidden / categories filter goes to the device as an admissibility mask,
//     so there are no over-fetch rounds.  Vectors are stored as given (GetVectors returns them unchanged) and indexed
//     with their entries sorted by index; a repeated index is rejected.
//   * Quantized collections are ErrNotSupported.
// Errors are the reference's sentinels (storage/errors.go:20-24) as exception types; all methods are serialised by one
// mutex per database (the reference requires goroutine safety; a GPU handle must be used by one caller at a time).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gorse_hip.h"
#include "gob.hpp"

namespace gorse {
namespace storage {
struct ErrNotFound : std::runtime_error { using std::runtime_error::runtime_error; };        // errors.go:20
struct ErrNotSupported : std::runtime_error { using std::runtime_error::runtime_error; };    // errors.go:23
struct ErrAlreadyExists : std::runtime_error { using std::runtime_error::runtime_error; };   // errors.go:24
}  // namespace storage

namespace vectors {

enum Distance { Cosine = 0, Euclidean = 1, Dot = 2 };  // database.go:29-33

struct VectorConfig {   // database.go:75-80
    std::string Type;   // "" | sq | pq | rq
    int Bits = 0;
};
struct CollectionInfo {  // database.go:83-88
    std::string Name;
    int Dimension = 0;
    Distance Dist = Cosine;
    VectorConfig Config;
};
struct Vector {  // database.go:90-97
    std::string Id;
    std::vector<float> Values;
    std::vector<uint32_t> Indices;
    bool IsHidden = false;
    std::vector<std::string> Categories;
    int64_t TimestampMs = 0;  // time.Time at the millisecond resolution the backends store
};
struct ScoredVector : Vector {  // database.go:100-103
    float Score = 0;
};

// Exact k-nearest search of one query over a dense row-major matrix; the default implementation is the GPU
// (HipSearcher), the CPU test-suite injects a checker built on the oracle.
struct Searcher {
    virtual ~Searcher() = default;
    // X changed since the last call (rows added / removed): drop any cached index of this collection
    virtual void invalidate(const std::string &collection) = 0;
    // for each of the nq queries (row-major Q): the k smallest distances, ascending, into idx / dist (nq x k) and how
    // many of them into cnt (nq)
    virtual void search(const std::string &collection, const float *X, int64_t n, int d, int metric, const float *Q,
                        int64_t nq, int k, int32_t *idx, float *dist, int32_t *cnt) = 0;
    // the same over the rows with admissible[r] != 0 only (hidden vectors, the categories filter), in ONE search; false =
    // not offered, the caller filters an over-fetched search instead
    virtual bool search_masked(const std::string &, const float *, int64_t, int, int, const uint8_t *, const float *, int64_t, int,
                               int32_t *, float *, int32_t *) {
        return false;
    }
};

// One gorse_topk handle per collection, rebuilt lazily after a change (like BruteforceHIP.sync in INTEGRATION.md).
class HipSearcher : public Searcher {
public:
    explicit HipSearcher(int device = 0) : device_(device) {}
    ~HipSearcher() override {
        for (auto &kv : handles_) gorse_topk_destroy(kv.second);
    }
    void invalidate(const std::string &collection) override {
        auto it = handles_.find(collection);
        if (it != handles_.end()) {
            gorse_topk_destroy(it->second);
            handles_.erase(it);
        }
    }
    void search(const std::string &collection, const float *X, int64_t n, int d, int metric, const float *Q, int64_t nq,
                int k, int32_t *idx, float *dist, int32_t *cnt) override {
        gorse_topk *&h = handles_[collection];
        if (!h) {
            if (gorse_topk_create(&h, device_, n, d, GORSE_DTYPE_F32, metric, X) != GORSE_OK) {
                handles_.erase(collection);
                throw std::runtime_error(std::string("gorse_topk_create: ") + gorse_hip_last_error());
            }
        }
        // one call for all queries: >= 768 of them run on the MFMA path of the library, fewer on its scan
        if (gorse_topk_search_vector(h, Q, nq, k, 0, idx, dist, cnt) != GORSE_OK)
            throw std::runtime_error(std::string("gorse_topk_search_vector: ") + gorse_hip_last_error());
    }
    // the filter as a device mask (gorse_topk_set_mask): the sweep never sees an inadmissible row, k stays topK
    bool search_masked(const std::string &collection, const float *X, int64_t n, int d, int metric, const uint8_t *admissible,
                       const float *Q, int64_t nq, int k, int32_t *idx, float *dist, int32_t *cnt) override {
        gorse_topk *&h = handles_[collection];
        if (!h) {
            if (gorse_topk_create(&h, device_, n, d, GORSE_DTYPE_F32, metric, X) != GORSE_OK) {
                handles_.erase(collection);
                throw std::runtime_error(std::string("gorse_topk_create: ") + gorse_hip_last_error());
            }
        }
        if (gorse_topk_set_mask(h, admissible) != GORSE_OK)
            throw std::runtime_error(std::string("gorse_topk_set_mask: ") + gorse_hip_last_error());
        const int32_t rc = gorse_topk_search_vector(h, Q, nq, k, 0, idx, dist, cnt);
        (void)gorse_topk_set_mask(h, nullptr);
        if (rc != GORSE_OK) throw std::runtime_error(std::string("gorse_topk_search_vector: ") + gorse_hip_last_error());
        return true;
    }

private:
    int device_;
    std::map<std::string, gorse_topk *> handles_;
};

// Exact top-k of sparse queries over sparse rows (CSR, indices strictly ascending per row), rows with admissible[r] == 0
// left out: the reference's QueryVectors on a sparse collection.  Default = the GPU, the CPU test-suite injects the oracle.
struct SparseSearcher {
    virtual ~SparseSearcher() = default;
    virtual void invalidate(const std::string &collection) = 0;
    virtual void search(const std::string &collection, int64_t n, const int64_t *indptr, const uint32_t *indices,
                        const float *values, const uint8_t *admissible, int64_t nq, const int64_t *q_indptr,
                        const uint32_t *q_indices, const float *q_values, int k, int32_t *idx, float *score,
                        int32_t *cnt) = 0;
};

class HipSparseSearcher : public SparseSearcher {  // one gorse_sparse handle per collection, rebuilt after a change
public:
    explicit HipSparseSearcher(int device = 0) : device_(device) {}
    ~HipSparseSearcher() override {
        for (auto &kv : handles_) gorse_sparse_destroy(kv.second);
    }
    void invalidate(const std::string &collection) override {
        auto it = handles_.find(collection);
        if (it != handles_.end()) {
            gorse_sparse_destroy(it->second);
            handles_.erase(it);
        }
    }
    void search(const std::string &collection, int64_t n, const int64_t *indptr, const uint32_t *indices, const float *values,
                const uint8_t *admissible, int64_t nq, const int64_t *q_indptr, const uint32_t *q_indices,
                const float *q_values, int k, int32_t *idx, float *score, int32_t *cnt) override {
        gorse_sparse *&h = handles_[collection];
        if (!h) {
            if (gorse_sparse_create(&h, device_, n, indptr, indices, values) != GORSE_OK) {
                handles_.erase(collection);
                throw std::runtime_error(std::string("gorse_sparse_create: ") + gorse_hip_last_error());
            }
        }
        if (gorse_sparse_set_mask(h, admissible) != GORSE_OK)
            throw std::runtime_error(std::string("gorse_sparse_set_mask: ") + gorse_hip_last_error());
        if (gorse_sparse_search(h, nq, q_indptr, q_indices, q_values, nullptr, k, idx, score, cnt) != GORSE_OK)
            throw std::runtime_error(std::string("gorse_sparse_search: ") + gorse_hip_last_error());
    }

private:
    int device_;
    std::map<std::string, gorse_sparse *> handles_;
};

class HipDatabase {
public:
    explicit HipDatabase(std::shared_ptr<Searcher> searcher = nullptr, std::shared_ptr<SparseSearcher> sparse = nullptr)
        : searcher_(searcher ? std::move(searcher) : std::make_shared<HipSearcher>()),
          sparse_(sparse ? std::move(sparse) : std::make_shared<HipSparseSearcher>()) {}

    void Init() {}
    void Optimize(const std::string &) {}
    void Close() {
        std::lock_guard<std::mutex> g(mu_);
        for (auto &kv : collections_) drop_index(kv.first);
        closed_ = true;
    }

    std::vector<std::string> ListCollections() {
        std::lock_guard<std::mutex> g(mu_);
        check_open();
        std::vector<std::string> names;
        for (auto &kv : collections_) names.push_back(kv.first);
        return names;  // std::map: sorted, like the sorted listing of the file-backed backends
    }
    CollectionInfo DescribeCollection(const std::string &name) {
        std::lock_guard<std::mutex> g(mu_);
        return coll(name).info;
    }
    void AddCollection(const std::string &name, int dimensions, Distance distance, const VectorConfig &config = {}) {
        std::lock_guard<std::mutex> g(mu_);
        check_open();
        if (dimensions < 0) throw std::invalid_argument("invalid vector dimension " + std::to_string(dimensions));
        if (!config.Type.empty()) throw storage::ErrNotSupported("quantization type " + config.Type + " for hip not supported");
        if (distance != Cosine && distance != Euclidean && distance != Dot) throw storage::ErrNotSupported("distance method not supported");
        if (dimensions == 0 && distance != Dot)  // xvec.go:243-245
            throw storage::ErrNotSupported("distance method for sparse vector not supported");
        if (collections_.count(name)) throw storage::ErrAlreadyExists("collection " + name + " already exists");
        Collection c;
        c.info.Name = name;
        c.info.Dimension = dimensions;
        c.info.Dist = distance;
        c.info.Config = config;
        collections_.emplace(name, std::move(c));
    }
    void DeleteCollection(const std::string &name) {
        std::lock_guard<std::mutex> g(mu_);
        check_open();
        if (!collections_.erase(name)) throw storage::ErrNotFound("collection " + name + ": not found");
        drop_index(name);
    }
    int64_t CountVectors(const std::string &name) {
        std::lock_guard<std::mutex> g(mu_);
        return (int64_t)coll(name).rows.size();
    }
    void AddVectors(const std::string &name, const std::vector<Vector> &vs) {
        std::lock_guard<std::mutex> g(mu_);
        if (vs.empty()) return;
        Collection &c = coll(name);
        const bool sparse = c.info.Dimension == 0;
        for (const Vector &v : vs) {  // validate everything before touching the collection (xvec.go:318-321)
            if (sparse) {
                if (v.Indices.empty() || v.Indices.size() != v.Values.size())
                    throw std::invalid_argument("vector " + v.Id + " is not a sparse vector (Indices and Values of one length)");
                std::vector<uint32_t> sorted(v.Indices);
                std::sort(sorted.begin(), sorted.end());
                if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end())
                    throw std::invalid_argument("vector " + v.Id + " repeats an index");
            } else if (!v.Indices.empty() || (int)v.Values.size() != c.info.Dimension) {
                throw std::invalid_argument("vector " + v.Id + " has dimension " + std::to_string(v.Values.size()) +
                                            (v.Indices.empty() ? "" : " (sparse)") + ", collection " + name + " has " +
                                            std::to_string(c.info.Dimension));
            }
        }
        for (const Vector &v : vs) {
            auto it = c.by_id.find(v.Id);
            if (it == c.by_id.end()) {
                c.by_id[v.Id] = c.rows.size();
                c.rows.push_back(v);
                if (!sparse) c.data.insert(c.data.end(), v.Values.begin(), v.Values.end());
            } else {  // upsert
                c.rows[it->second] = v;
                if (!sparse) std::copy(v.Values.begin(), v.Values.end(), c.data.begin() + it->second * c.info.Dimension);
            }
        }
        c.csr_valid = false;
        drop_index(name);
    }
    std::vector<Vector> GetVectors(const std::string &name, const std::vector<std::string> &ids) {
        std::lock_guard<std::mutex> g(mu_);
        Collection &c = coll(name);
        std::vector<Vector> out;
        std::map<std::string, bool> seen;
        for (const std::string &id : ids) {
            if (seen.count(id)) continue;
            seen[id] = true;
            auto it = c.by_id.find(id);
            if (it != c.by_id.end()) out.push_back(c.rows[it->second]);
        }
        return out;
    }
    void DeleteVectors(const std::string &name, int64_t timestamp_ms) {
        std::lock_guard<std::mutex> g(mu_);
        Collection &c = coll(name);
        std::vector<Vector> keep;
        for (Vector &v : c.rows)
            if (!(v.TimestampMs < timestamp_ms)) keep.push_back(std::move(v));
        if (keep.size() == c.rows.size()) {
            c.rows = std::move(keep);
            return;
        }
        c.rows = std::move(keep);
        c.by_id.clear();
        c.data.clear();
        for (size_t r = 0; r < c.rows.size(); r++) {
            c.by_id[c.rows[r].Id] = r;
            if (c.info.Dimension != 0) c.data.insert(c.data.end(), c.rows[r].Values.begin(), c.rows[r].Values.end());
        }
        c.csr_valid = false;
        drop_index(name);
    }
    std::vector<ScoredVector> QueryVectors(const std::string &name, const Vector &q, const std::vector<std::string> &categories,
                                           int topK) {
        if (!q.Indices.empty()) {
            auto r = QuerySparseBatch(name, {q}, categories, topK);
            return r.empty() ? std::vector<ScoredVector>() : std::move(r[0]);
        }
        auto r = QueryVectorsBatch(name, q.Values, 1, categories, topK);
        return r.empty() ? std::vector<ScoredVector>() : std::move(r[0]);
    }
    // Sparse queries (Indices / Values) against a sparse collection, all in ONE device search: every admissible vector
    // ranked by its inner product with the query, cut to topK, zero scores dropped (xvec.go:379-446).
    std::vector<std::vector<ScoredVector>> QuerySparseBatch(const std::string &name, const std::vector<Vector> &queries,
                                                            const std::vector<std::string> &categories, int topK) {
        std::lock_guard<std::mutex> g(mu_);
        Collection &c = coll(name);
        std::vector<std::vector<ScoredVector>> out(queries.size());
        if (topK <= 0 || queries.empty()) return out;
        if (c.info.Dimension != 0) throw std::invalid_argument("sparse query against the dense collection " + name);
        if (topK > 1024) throw storage::ErrNotSupported("topK > 1024 on a sparse collection for hip not supported");
        std::vector<int64_t> qp{0};
        std::vector<uint32_t> qi;
        std::vector<float> qv;
        for (const Vector &q : queries) {
            if (q.Indices.size() != q.Values.size()) throw std::invalid_argument("sparse query: Indices and Values differ in length");
            const size_t before = qi.size();
            append_sorted(q, qi, qv);
            if (std::adjacent_find(qi.begin() + (std::ptrdiff_t)before, qi.end()) != qi.end())
                throw std::invalid_argument("sparse query repeats an index");
            qp.push_back((int64_t)qi.size());
        }
        const int64_t n = (int64_t)c.rows.size();
        if (n == 0) return out;
        if (!c.csr_valid) {
            c.indptr.assign(1, 0);
            c.indices.clear();
            c.values.clear();
            for (const Vector &v : c.rows) {
                append_sorted(v, c.indices, c.values);
                c.indptr.push_back((int64_t)c.indices.size());
            }
            c.csr_valid = true;
        }
        std::vector<uint8_t> ok((size_t)n);
        for (int64_t r = 0; r < n; r++) ok[(size_t)r] = admissible(c.rows[(size_t)r], categories);
        const int64_t nq = (int64_t)queries.size();
        std::vector<int32_t> idx((size_t)(nq * topK), -1), cnt((size_t)nq, 0);
        std::vector<float> score((size_t)(nq * topK), 0.0f);
        sparse_->search(name, n, c.indptr.data(), c.indices.data(), c.values.data(), ok.data(), nq, qp.data(), qi.data(),
                        qv.data(), topK, idx.data(), score.data(), cnt.data());
        for (int64_t t = 0; t < nq; t++)
            for (int e = 0; e < cnt[(size_t)t]; e++) {
                ScoredVector sv;
                static_cast<Vector &>(sv) = c.rows[(size_t)idx[(size_t)(t * topK + e)]];
                sv.Score = score[(size_t)(t * topK + e)];  // Dot: the inner product itself
                out[(size_t)t].push_back(std::move(sv));
            }
        return out;
    }
    // The bulk form: nq dense queries (row-major) against one collection with one filter, in ONE
    // device search per over-fetch round -- what replaces the per-user QueryVectors loop of worker/pipeline.go:403-448.
    std::vector<std::vector<ScoredVector>> QueryVectorsBatch(const std::string &name, const std::vector<float> &queries,
                                                             int64_t nq, const std::vector<std::string> &categories,
                                                             int topK) {
        std::lock_guard<std::mutex> g(mu_);
        Collection &c = coll(name);
        std::vector<std::vector<ScoredVector>> out((size_t)std::max<int64_t>(nq, 0));
        if (topK <= 0 || nq <= 0) return out;
        const int d = c.info.Dimension;
        if (d == 0) throw std::invalid_argument("dense query against the sparse collection " + name);
        if ((int64_t)queries.size() != nq * d)
            throw std::invalid_argument("query has dimension " + std::to_string(nq ? queries.size() / nq : 0) + ", collection " +
                                        name + " has " + std::to_string(d));
        const int64_t n = (int64_t)c.rows.size();
        if (n == 0) return out;
        const int metric = c.info.Dist == Dot ? GORSE_METRIC_NEG_DOT : (c.info.Dist == Euclidean ? GORSE_METRIC_EUCLIDEAN : GORSE_METRIC_COSINE);
        std::vector<uint8_t> ok((size_t)n);
        bool filtered = false;
        for (int64_t r = 0; r < n; r++) {
            ok[(size_t)r] = admissible(c.rows[(size_t)r], categories);
            filtered = filtered || !ok[(size_t)r];
        }
        if (filtered) {  // the filter as a device mask, where the searcher offers it: one search with k = topK
            const int64_t k1 = std::min<int64_t>(n, topK);
            std::vector<int32_t> idx1((size_t)(nq * k1), -1), cnt1((size_t)nq, 0);
            std::vector<float> dist1((size_t)(nq * k1), 0.0f);
            if (searcher_->search_masked(name, c.data.data(), n, d, metric, ok.data(), queries.data(), nq, (int)k1, idx1.data(),
                                         dist1.data(), cnt1.data())) {
                for (int64_t t = 0; t < nq; t++)
                    for (int e = 0; e < cnt1[(size_t)t]; e++) {
                        ScoredVector s;
                        static_cast<Vector &>(s) = c.rows[(size_t)idx1[(size_t)(t * k1 + e)]];
                        s.Score = -dist1[(size_t)(t * k1 + e)];
                        out[(size_t)t].push_back(std::move(s));
                    }
                return out;
            }
        }
        // queries still short of topK admissible vectors are searched again with a 4x larger k
        std::vector<int64_t> todo((size_t)nq);
        for (int64_t t = 0; t < nq; t++) todo[(size_t)t] = t;
        int64_t k = std::min<int64_t>(n, std::max<int64_t>(2 * (int64_t)topK, (int64_t)topK + 32));
        std::vector<float> Q;
        std::vector<int32_t> idx, cnt;
        std::vector<float> dist;
        while (!todo.empty()) {
            const int64_t m = (int64_t)todo.size();
            const float *qp = queries.data();
            if (m != nq) {
                Q.resize((size_t)m * d);
                for (int64_t t = 0; t < m; t++)
                    std::copy(queries.begin() + todo[(size_t)t] * d, queries.begin() + (todo[(size_t)t] + 1) * d, Q.begin() + t * d);
                qp = Q.data();
            }
            idx.assign((size_t)(m * k), -1);
            dist.assign((size_t)(m * k), 0.0f);
            cnt.assign((size_t)m, 0);
            searcher_->search(name, c.data.data(), n, d, metric, qp, m, (int)k, idx.data(), dist.data(), cnt.data());
            std::vector<int64_t> again;
            for (int64_t t = 0; t < m; t++) {
                std::vector<ScoredVector> &res = out[(size_t)todo[(size_t)t]];
                res.clear();
                for (int e = 0; e < cnt[(size_t)t] && (int)res.size() < topK; e++) {
                    const int32_t r = idx[(size_t)(t * k + e)];
                    if (!ok[(size_t)r]) continue;
                    ScoredVector s;
                    static_cast<Vector &>(s) = c.rows[(size_t)r];
                    s.Score = -dist[(size_t)(t * k + e)];  // Dot: a.b; Euclidean / Cosine: negated distance (xvec.go:425-427)
                    res.push_back(std::move(s));
                }
                if ((int)res.size() < topK && k < n) again.push_back(todo[(size_t)t]);
            }
            if (k >= n) break;
            todo.swap(again);
            k = std::min<int64_t>(n, k * 4);
        }
        return out;
    }

private:
    struct Collection {
        CollectionInfo info;
        std::vector<Vector> rows;               // insertion order; row r of `data`
        std::vector<float> data;                // dense: rows x Dimension, row-major: what the searcher indexes
        std::map<std::string, size_t> by_id;
        // sparse: the rows as CSR with ascending indices, rebuilt lazily after a change
        bool csr_valid = false;
        std::vector<int64_t> indptr;
        std::vector<uint32_t> indices;
        std::vector<float> values;
    };
    // hidden vectors never match; `categories` is CONTAIN_ALL (xvec.go:386-394)
    static bool admissible(const Vector &v, const std::vector<std::string> &categories) {
        bool a = !v.IsHidden;
        for (const std::string &cat : categories)
            a = a && std::find(v.Categories.begin(), v.Categories.end(), cat) != v.Categories.end();
        return a;
    }
    // (index, value) pairs of v appended in ascending index order
    static void append_sorted(const Vector &v, std::vector<uint32_t> &indices, std::vector<float> &values) {
        std::vector<size_t> order(v.Indices.size());
        for (size_t t = 0; t < order.size(); t++) order[t] = t;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return v.Indices[a] < v.Indices[b]; });
        for (size_t t : order) {
            indices.push_back(v.Indices[t]);
            values.push_back(v.Values[t]);
        }
    }
    void drop_index(const std::string &name) {
        searcher_->invalidate(name);
        sparse_->invalidate(name);
    }
    void check_open() const {
        if (closed_) throw std::runtime_error("hip vector database is closed");
    }
    Collection &coll(const std::string &name) {
        check_open();
        auto it = collections_.find(name);
        if (it == collections_.end()) throw storage::ErrNotFound("collection " + name + ": not found");
        return it->second;
    }
    std::mutex mu_;
    bool closed_ = false;
    std::shared_ptr<Searcher> searcher_;
    std::shared_ptr<SparseSearcher> sparse_;
    std::map<std::string, Collection> collections_;
};

// vectors.Register / vectors.Open (database.go:155-175): creators by URL prefix
using Creator = std::function<std::shared_ptr<HipDatabase>(const std::string &path, const std::string &tablePrefix)>;
inline std::map<std::string, Creator> &creators() {
    static std::map<std::string, Creator> c = {
        {"hip://", [](const std::string &, const std::string &) { return std::make_shared<HipDatabase>(); }}};
    return c;
}
inline void Register(const std::vector<std::string> &prefixes, Creator creator) {
    for (const std::string &p : prefixes) creators()[p] = creator;
}
inline std::shared_ptr<HipDatabase> Open(const std::string &path, const std::string &tablePrefix = "") {
    for (auto &kv : creators())
        if (path.compare(0, kv.first.size(), kv.first) == 0) return kv.second(path, tablePrefix);
    throw std::runtime_error("Unknown database: " + path);
}

inline std::string ItemToItemCollection(const std::string &name) { return "item_to_item_" + name; }   // database.go:56-58
inline std::string UserToUserCollection(const std::string &name) { return "user_to_user_" + name; }   // database.go:60-62

}  // namespace vectors

// ---- logics: the similarity recommenders over a vectors.Database ------------------------------------------------------
// logics/vector_writer.go:33-209 (VectorWriter, newSparseVector), logics/item_to_item.go:50-245 (QueryItemToItem and the
// embedding / tags / users / auto kinds), logics/user_to_user.go:50-237 (the same with users: embedding / tags / items /
// auto).  The sparse kinds store sqrt(idf)-weighted id sets in a sparse Dot collection, so that the inner product of two
// vectors is the sum of the idf of their common ids.  Column expressions (expr programs over data.Item / data.User) are
// the caller's: the twins take the extracted embedding / label ids.
namespace logics {

struct Score {  // cache.Score as QueryItemToItem fills it (item_to_item.go:81)
    std::string Id;
    double Value = 0;
    std::vector<std::string> Categories;
};

class VectorWriter {  // collection validation + batched AddVectors (vector_writer.go:35-190); sparse: dimension 0
public:
    VectorWriter(std::shared_ptr<vectors::HipDatabase> client, std::string collection, vectors::Distance distance,
                 int64_t timestamp_ms, int batchSize = 0, bool sparse = false)
        : client_(std::move(client)), collection_(std::move(collection)), distance_(distance), timestamp_(timestamp_ms),
          batch_(batchSize > 0 ? batchSize : 1024), sparse_(sparse), dimension_(sparse ? 0 : -1) {}
    void Add(const vectors::Vector &v) {
        std::lock_guard<std::mutex> g(mu_);
        if (sparse_) {  // vector_writer.go:88-91: silently skipped
            if (v.Indices.empty() || v.Indices.size() != v.Values.size()) return;
        } else if (!v.Indices.empty() || v.Values.empty()) {  // :92-96
            return;
        }
        buffer_.push_back(v);
        if ((int)buffer_.size() >= batch_) flush();
    }
    void Clean() {  // flush, then drop what an earlier refresh left behind (vector_writer.go:104-114)
        std::lock_guard<std::mutex> g(mu_);
        flush();
        try {
            client_->DeleteVectors(collection_, timestamp_);
        } catch (const storage::ErrNotFound &) {
        }
    }

private:
    void flush() {
        if (buffer_.empty()) return;
        if (sparse_) {  // no dimension bookkeeping (vector_writer.go:152: only `if !w.sparse`)
            std::vector<vectors::Vector> all;
            all.swap(buffer_);
            ensure_collection();
            client_->AddVectors(collection_, all);
            return;
        }
        if (dimension_ < 0) {  // the most frequent dimension of the first batch, first seen wins ties (:153-166)
            std::map<size_t, int> counts;
            for (auto &v : buffer_) counts[v.Values.size()]++;
            size_t dim = buffer_[0].Values.size();
            for (auto &v : buffer_)
                if (counts[v.Values.size()] > counts[dim]) dim = v.Values.size();
            dimension_ = (int)dim;
        }
        std::vector<vectors::Vector> ok;
        for (auto &v : buffer_)
            if ((int)v.Values.size() == dimension_) ok.push_back(v);  // others are logged and dropped (:169-180)
        buffer_.clear();
        ensure_collection();
        client_->AddVectors(collection_, ok);
    }
    void ensure_collection() {  // vector_writer.go:116-147: recreate when dimension / distance / config differ
        if (exists_) return;
        bool have = false;
        try {
            auto info = client_->DescribeCollection(collection_);
            have = true;
            if (info.Dimension != dimension_ || info.Dist != distance_ || !info.Config.Type.empty() || info.Config.Bits != 0) {
                try {
                    client_->DeleteCollection(collection_);
                } catch (const storage::ErrNotFound &) {
                }
                have = false;
            }
        } catch (const storage::ErrNotFound &) {
        }
        if (!have) client_->AddCollection(collection_, dimension_, distance_);
        exists_ = true;
    }
    std::shared_ptr<vectors::HipDatabase> client_;
    std::string collection_;
    vectors::Distance distance_;
    int64_t timestamp_;
    int batch_;
    bool sparse_;
    std::mutex mu_;
    int dimension_;
    bool exists_ = false;
    std::vector<vectors::Vector> buffer_;
};

// newSparseVector / appendSparseVector (vector_writer.go:192-209): ids outside the idf table or with idf <= 0 are
// dropped, the value is float32(math.Sqrt(float64(idf))), the index is offset + id
inline void appendSparseVector(vectors::Vector &v, const std::vector<int32_t> &ids, const std::vector<float> &idf, uint32_t offset) {
    for (int32_t id : ids) {
        if (id < 0 || (size_t)id >= idf.size() || !(idf[(size_t)id] > 0)) continue;
        v.Indices.push_back(offset + (uint32_t)id);
        v.Values.push_back((float)std::sqrt((double)idf[(size_t)id]));
    }
}
// a set of label ids in ascending order: mapset + slices.Sort (item_to_item.go:187-191)
inline std::vector<int32_t> sorted_set(std::vector<int32_t> ids) {
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    return ids;
}
// feedback ids in ascending order: slices.Sort (item_to_item.go:213).  A repeated id -- the reference would hand xvec a
// vector with a repeated index, whose treatment no test pins -- is kept once.
inline std::vector<int32_t> sorted_feedback(std::vector<int32_t> ids) { return sorted_set(std::move(ids)); }

// The three sparse kinds of NewItemToItem / NewUserToUser (item_to_item.go:89-113, user_to_user.go:89-113).  `meta` carries
// Id / IsHidden / Categories (users have neither of the last two); Timestamp is the writer's.
inline vectors::Vector tagsVector(vectors::Vector meta, const std::vector<int32_t> &tags, const std::vector<float> &tagsIDF) {
    appendSparseVector(meta, sorted_set(tags), tagsIDF, 0);  // tagsItemToItem.Add, item_to_item.go:179-199
    return meta;
}
inline vectors::Vector feedbackVector(vectors::Vector meta, const std::vector<int32_t> &feedback, const std::vector<float> &idf) {
    appendSparseVector(meta, sorted_feedback(feedback), idf, 0);  // usersItemToItem.Add, :212-220
    return meta;
}
inline vectors::Vector autoVector(vectors::Vector meta, const std::vector<int32_t> &tags, const std::vector<float> &tagsIDF,
                                  const std::vector<int32_t> &feedback, const std::vector<float> &feedbackIDF) {
    appendSparseVector(meta, sorted_set(tags), tagsIDF, 0);  // autoItemToItem.Add, :232-245: feedback ids after the tags
    appendSparseVector(meta, sorted_feedback(feedback), feedbackIDF, (uint32_t)tagsIDF.size());
    return meta;
}

// neighbour list -> scores, the loop of item_to_item.go:72-86 for the embedding type (distance Euclidean, scale 1):
// the item itself is skipped, score = 1 / (1 - Score) = 1 / (1 + distance), at most n entries
inline std::vector<Score> embedding_scores(const std::vector<vectors::ScoredVector> &neighbors, const std::string &self, int n) {
    std::vector<Score> out;
    for (const auto &nb : neighbors) {
        if (nb.Id == self) continue;
        Score s;
        s.Id = nb.Id;
        s.Value = 1.0 / (1.0 - (double)nb.Score);
        s.Categories = nb.Categories;
        out.push_back(std::move(s));
        if ((int)out.size() == n) break;
    }
    return out;
}

// neighbour list -> scores for ANY kind (item_to_item.go:64-87): distance Dot unless the kind is "embedding"; the item
// itself and, for Dot, scores <= 0 are skipped; "auto" halves the score (tags + feedback both contribute)
inline std::vector<Score> similar_scores(const std::vector<vectors::ScoredVector> &neighbors, const std::string &type,
                                         const std::string &self, int n) {
    if (type == "embedding") return embedding_scores(neighbors, self, n);
    const double scale = type == "auto" ? .5 : 1.0;
    std::vector<Score> out;
    for (const auto &nb : neighbors) {
        if (nb.Id == self || nb.Score <= 0) continue;
        Score s;
        s.Id = nb.Id;
        s.Value = (double)nb.Score * scale;
        s.Categories = nb.Categories;
        out.push_back(std::move(s));
        if ((int)out.size() == n) break;
    }
    return out;
}
// QueryItemToItem / QueryUserToUser for any kind: GetVectors(id), QueryVectors(n + 1), similar_scores
inline std::vector<Score> QuerySimilarTyped(vectors::HipDatabase &client, const std::string &collection, const std::string &type,
                                            const std::string &id, const std::vector<std::string> &categories, int n) {
    auto queries = client.GetVectors(collection, {id});
    if (queries.empty()) return {};
    return similar_scores(client.QueryVectors(collection, queries[0], categories, n + 1), type, id, n);
}
// the refresh of a sparse kind for many ids with ONE device search (HipDatabase::QuerySparseBatch)
inline std::vector<std::vector<Score>> QuerySimilarTypedBulk(vectors::HipDatabase &client, const std::string &collection,
                                                             const std::string &type, const std::vector<std::string> &ids,
                                                             const std::vector<std::string> &categories, int n) {
    std::vector<std::vector<Score>> out(ids.size());
    std::vector<vectors::Vector> queries;
    std::vector<size_t> which;
    for (size_t t = 0; t < ids.size(); t++) {
        auto v = client.GetVectors(collection, {ids[t]});
        if (v.empty()) continue;
        queries.push_back(std::move(v[0]));
        which.push_back(t);
    }
    if (which.empty()) return out;
    auto res = client.QuerySparseBatch(collection, queries, categories, n + 1);
    for (size_t r = 0; r < which.size(); r++) out[which[r]] = similar_scores(res[r], type, ids[which[r]], n);
    return out;
}

// QueryItemToItem / QueryUserToUser for Type == "embedding" (item_to_item.go:50-88): GetVectors(id), QueryVectors(n + 1)
inline std::vector<Score> QuerySimilar(vectors::HipDatabase &client, const std::string &collection, const std::string &id,
                                       const std::vector<std::string> &categories, int n) {
    auto queries = client.GetVectors(collection, {id});
    if (queries.empty()) return {};
    return embedding_scores(client.QueryVectors(collection, queries[0], categories, n + 1), id, n);
}

// The same for many ids with ONE bulk search (vectors::HipDatabase::QueryVectorsBatch): the refresh loop of
// master/tasks.go:930-962 / worker/pipeline.go:403-448 asks for every item's neighbours, which is the all-pairs top-k.
inline std::vector<std::vector<Score>> QuerySimilarBulk(vectors::HipDatabase &client, const std::string &collection,
                                                        const std::vector<std::string> &ids,
                                                        const std::vector<std::string> &categories, int n) {
    std::vector<std::vector<Score>> out(ids.size());
    std::vector<float> Q;
    std::vector<size_t> which;
    int d = 0;
    for (size_t t = 0; t < ids.size(); t++) {
        auto v = client.GetVectors(collection, {ids[t]});
        if (v.empty()) continue;  // unknown id: empty list, like the nil of item_to_item.go:56-58
        d = (int)v[0].Values.size();
        Q.insert(Q.end(), v[0].Values.begin(), v[0].Values.end());
        which.push_back(t);
    }
    if (which.empty()) return out;
    auto res = client.QueryVectorsBatch(collection, Q, (int64_t)which.size(), categories, n + 1);
    (void)d;
    for (size_t r = 0; r < which.size(); r++) out[which[r]] = embedding_scores(res[r], ids[which[r]], n);
    return out;
}

// worker/pipeline.go:403-425 (updateCollaborativeRecommend) for MANY users at once: the reference asks the vector
// database for CacheSize + |excludeSet| neighbours of the user's embedding in the collaborative_filtering_<id> collection
// (distance Dot, filled by master/tasks.go:930-962) and drops the excluded ids.  Here all users go through one bulk
// search with k = CacheSize + the largest exclude set; each user's list is the first CacheSize + |exclude_u| entries of
// its row with the excluded ids dropped -- the per-user result of the reference (ties at a cut may order differently).
struct UserQuery {
    std::vector<float> Embedding;
    std::vector<std::string> Exclude;
};
inline std::vector<std::vector<Score>> CollaborativeRecommendBulk(vectors::HipDatabase &client, const std::string &collection,
                                                                  const std::vector<UserQuery> &users, int cacheSize) {
    std::vector<std::vector<Score>> out(users.size());
    if (users.empty()) return out;
    size_t max_ex = 0;
    std::vector<float> Q;
    for (const auto &u : users) {
        max_ex = std::max(max_ex, u.Exclude.size());
        Q.insert(Q.end(), u.Embedding.begin(), u.Embedding.end());
    }
    auto res = client.QueryVectorsBatch(collection, Q, (int64_t)users.size(), {}, cacheSize + (int)max_ex);
    for (size_t t = 0; t < users.size(); t++) {
        const size_t take = (size_t)cacheSize + users[t].Exclude.size();
        for (size_t e = 0; e < res[t].size() && e < take; e++) {
            const auto &v = res[t][e];
            if (std::find(users[t].Exclude.begin(), users[t].Exclude.end(), v.Id) != users[t].Exclude.end()) continue;
            Score sc;
            sc.Id = v.Id;
            sc.Value = (double)v.Score;
            sc.Categories = v.Categories;
            out[t].push_back(std::move(sc));
        }
    }
    return out;
}

// MatrixFactorizationUsers (logics/cf.go:122-179): user id -> embedding, and its blob: WriteGob(int64 count), then per
// user WriteString(id) (little-endian int32 length + bytes) + WriteSlice(embedding) (int32 length + little-endian float32s).
// The reference iterates a Go map (any order); here ids go out sorted.
class MatrixFactorizationUsers {
public:
    void Add(const std::string &userId, std::vector<float> v) { embeddings_[userId] = std::move(v); }
    bool Get(const std::string &userId, std::vector<float> &out) const {
        auto it = embeddings_.find(userId);
        if (it == embeddings_.end()) return false;
        out = it->second;
        return true;
    }
    size_t Count() const { return embeddings_.size(); }
    std::string Marshal() const {
        std::string w;
        put_bytes(w, gob::encode_int((int64_t)embeddings_.size()));
        for (const auto &kv : embeddings_) {
            put_bytes(w, kv.first);
            put_i32(w, (int32_t)kv.second.size());
            w.append((const char *)kv.second.data(), kv.second.size() * sizeof(float));
        }
        return w;
    }
    void Unmarshal(const std::string &blob) {
        size_t at = 0;
        const int64_t n = gob::decode_int(get_bytes(blob, at));
        embeddings_.clear();
        for (int64_t k = 0; k < n; k++) {
            std::string id = get_bytes(blob, at);
            const int32_t len = get_i32(blob, at);
            if (len < 0 || blob.size() - at < (size_t)len * sizeof(float)) throw std::runtime_error("unexpected EOF");
            std::vector<float> v((size_t)len);
            std::memcpy(v.data(), blob.data() + at, (size_t)len * sizeof(float));
            at += (size_t)len * sizeof(float);
            embeddings_[id] = std::move(v);
        }
    }

private:
    static void put_i32(std::string &w, int32_t v) { w.append((const char *)&v, 4); }
    static void put_bytes(std::string &w, const std::string &b) {  // encoding.WriteBytes
        put_i32(w, (int32_t)b.size());
        w += b;
    }
    static int32_t get_i32(const std::string &b, size_t &at) {
        if (b.size() - at < 4) throw std::runtime_error("unexpected EOF");
        int32_t v;
        std::memcpy(&v, b.data() + at, 4);
        at += 4;
        return v;
    }
    static std::string get_bytes(const std::string &b, size_t &at) {
        const int32_t n = get_i32(b, at);
        if (n < 0 || b.size() - at < (size_t)n) throw std::runtime_error("unexpected EOF");
        std::string out = b.substr(at, (size_t)n);
        at += (size_t)n;
        return out;
    }
    std::map<std::string, std::vector<float>> embeddings_;
};

// MatrixFactorizationItems (logics/cf.go:36-128): item id -> embedding with a nearest-item search by -floats.Dot, and its
// blob.  The reference keeps an ann.HNSW graph and writes it into the blob (cf.go:81-101 -> hnsw.go:278-337); here the
// neighbours come from the exact index, so there is no graph to write.  The blob keeps the reference's
// framing -- WriteGob(time.Time), WriteGob(int dimension), the index, WriteGob(int64 count), one WriteGob(string) per id --
// and only the index section differs.  Cross-build files:
//   * Unmarshal READS THE REFERENCE'S INDEX SECTION (HNSW.Marshal: float32 levelFactor, four int64 parameters, int64 count,
//     one WriteGob([]float32) per vector, count bottom neighbour queues, int64 layers x (int32 count x (int32 key, queue)),
//     int32 enterPoint; a queue = one bool + WriteSlice of 8-byte (int32, float32) pairs, common/heap/pq.go:128-133) and
//     keeps the vectors; the graph is skipped.  A master without this library can hand its file to a worker with it.
//   * Marshal writes an index section that a reference reader REJECTS CLEANLY: where HNSW.Unmarshal expects levelFactor it
//     finds the bytes "GHIP" (as a float32: 1.4e10, no level factor), then an int64 version (1) in the maxConnection slot,
//     three int64 zeros, the vector count 1 and an EMPTY gob stream (int32 0) for that "vector" -- encoding.ReadGob returns
//     gob's EOF, an ordinary error, before anything is allocated.  Behind it: int64 count, int32 dimension, the float32 rows.
class MatrixFactorizationItems {
public:
    explicit MatrixFactorizationItems(int64_t timestampUnixNanos = 0, std::shared_ptr<vectors::Searcher> searcher = nullptr)
        : timestamp_(timestampUnixNanos), searcher_(std::move(searcher)) {}
    void Add(const std::string &itemId, const std::vector<float> &v) {
        if (dimension_ == 0)
            dimension_ = (int)v.size();
        else if (dimension_ != (int)v.size())
            return;  // "dimension mismatch" is logged and the item dropped (cf.go:56-61)
        items_.push_back(itemId);
        data_.insert(data_.end(), v.begin(), v.end());
        if (searcher_) searcher_->invalidate(kCollection);
    }
    int64_t Timestamp() const { return timestamp_; }
    int Dimension() const { return dimension_; }
    size_t Count() const { return items_.size(); }
    const std::string &Id(size_t i) const { return items_[i]; }
    const float *Row(size_t i) const { return data_.data() + i * (size_t)dimension_; }
    // Search (cf.go:69-79): the n nearest items by -dot, Score = the inner product
    std::vector<Score> Search(const std::vector<float> &v, int n) {
        std::vector<Score> out;
        if (items_.empty() || n <= 0) return out;
        if ((int)v.size() != dimension_) throw std::invalid_argument("floats: slice lengths do not match");
        if (!searcher_) searcher_ = std::make_shared<vectors::HipSearcher>();
        std::vector<int32_t> idx((size_t)n);
        std::vector<float> dist((size_t)n);
        int32_t cnt = 0;
        searcher_->search(kCollection, data_.data(), (int64_t)items_.size(), dimension_, GORSE_METRIC_NEG_DOT, v.data(), 1, n,
                          idx.data(), dist.data(), &cnt);
        for (int32_t t = 0; t < cnt; t++) {
            Score sc;
            sc.Id = items_[(size_t)idx[(size_t)t]];
            sc.Value = -(double)dist[(size_t)t];
            out.push_back(std::move(sc));
        }
        return out;
    }
    std::string Marshal() const {
        std::string w;
        put_gob(w, gob::encode_time_unix_nanos(timestamp_));
        put_gob(w, gob::encode_int(dimension_));
        w += "GHIP";               // levelFactor slot
        put<int64_t>(w, 1);        // maxConnection slot: the version of this section
        put<int64_t>(w, 0);
        put<int64_t>(w, 0);
        put<int64_t>(w, 0);
        put<int64_t>(w, 1);        // "one vector" ...
        put<int32_t>(w, 0);        // ... whose gob stream is empty: the reference's reader stops here with an error
        put<int64_t>(w, (int64_t)items_.size());
        put<int32_t>(w, dimension_);
        w.append((const char *)data_.data(), data_.size() * sizeof(float));
        put_gob(w, gob::encode_int((int64_t)items_.size()));
        for (const auto &id : items_) put_gob(w, gob::encode_string(id));
        return w;
    }
    // The blob with its index section in the REFERENCE'S OWN format (HNSW.Marshal, hnsw.go:278-337) and a graph BUILT BY THE DEVICE,
    // for clusters in which some workers run a build without this library: their HNSW.Unmarshal loads the file and their own
    // knnSearch (hnsw.go:100-114) walks it.  The graph is what insert (hnsw.go:117-185) aims at, computed exactly instead of
    // searched for: every vector draws its level floor(-ln(u) * levelFactor) (hnsw.go:137); layer L holds the vectors of level
    // >= L, and a vector's neighbours in a layer are its nearest vectors OF THAT LAYER by -dot plus a few reverse links (write_layer)
    // -- maxConnection0 = 96 at the bottom, maxConnection = 48 above (NewHNSW's parameters, hnsw.go:52-60) -- from one exact
    // all-pairs search per layer on the device (a million vectors: one MFMA pass instead of a million efConstruction = 100
    // searches).  The levels come from one splitmix64 stream seeded by the count: the Go twin (integration/go/common/ann/
    // bruteforce_hip.go) draws the same stream, so both write the same blob for the same model.  Each queue is written ascending in
    // the distance, which is a valid heap array for heap.PriorityQueue (pq.go:42-48); the enter point is the first vector of the
    // top layer.  This library's own reader keeps the vectors of such a file and skips the graph, like any reference file.
    // The level of every vector of a MarshalReference stream: hnsw.go:137 (floor(-log(u) * levelFactor)) on ONE splitmix64 stream seeded
    // by the count.  Both twins use the SAME arithmetic, step for step -- the logarithm in double, rounded to float, times the float
    // factor, floored in double; levelFactor = float(1 / log(48.0)) in double -- because a float logf and a rounded double log can
    // differ by one ulp, and a product that straddles an integer would then give the Go twin (integration/go/common/ann/
    // bruteforce_hip.go: float32(math.Log(float64(u)))) and this one different levels, hence different blobs.
    // tests/test_items_blob_cpu.py compares the stream with a numpy restatement for a million draws.
    static float ReferenceLevelFactor() { return (float)(1.0 / std::log(48.0)); }
    static std::vector<int> ReferenceLevels(int64_t n, int *top_out = nullptr) {
        const float levelFactor = ReferenceLevelFactor();
        std::vector<int> level((size_t)std::max<int64_t>(n, 0), 0);
        uint64_t st = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
        int top = 0;
        for (int64_t i = 0; i < n; i++) {
            st += 0x9E3779B97F4A7C15ull;
            uint64_t z = st;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            const float u = ((float)(z >> 40) + 1.0f) / 16777216.0f;  // (0, 1]
            const float lg = (float)std::log((double)u);
            const float prod = -lg * levelFactor;  // one float multiplication, as the Go twin's
            level[(size_t)i] = (int)std::floor((double)prod);
            top = std::max(top, level[(size_t)i]);
        }
        if (top_out) *top_out = top;
        return level;
    }
    std::string MarshalReference() {
        constexpr int kM = 48, kM0 = 96, kEfConstruction = 100;
        const float levelFactor = ReferenceLevelFactor();
        const int64_t n = (int64_t)items_.size();
        const int d = dimension_;
        std::string w;
        put_gob(w, gob::encode_time_unix_nanos(timestamp_));
        put_gob(w, gob::encode_int(dimension_));
        put<float>(w, levelFactor);
        put<int64_t>(w, kM);
        put<int64_t>(w, kM0);
        put<int64_t>(w, 0);  // ef: efSearchValue falls back to efConstruction (hnsw.go:268-273)
        put<int64_t>(w, kEfConstruction);
        put<int64_t>(w, n);
        for (int64_t i = 0; i < n; i++) put_gob(w, gob::encode_f32_slice(Row((size_t)i), (size_t)d));
        // levels: one splitmix64 stream seeded by the count (the reference's rand.Float32 stream cannot be reproduced: SURVEY 8c)
        int top = 0;
        const std::vector<int> level = ReferenceLevels(n, &top);
        if (!searcher_) searcher_ = std::make_shared<vectors::HipSearcher>();
        // The neighbour queues of one layer: members (ascending ids; empty = every vector), at most cap neighbours each.
        // A queue = the vector's nearest cap - cap / 8 other members (exact) + up to cap / 8 REVERSE links: members that list this
        // vector among their nearest without being listed back, the ones with the fewest incoming links first.  Under -dot on
        // factors of unequal length every exact list points at the same long vectors, and a short vector that nobody lists could
        // never be reached by the reference's walk (insert links both ways, hnsw.go:163-183, and so keeps such vectors attached
        // while their neighbours' queues have room; its shrink keeps the nearest, which here would give the exact list back: a
        // reverse link is by construction farther than every exact neighbour).  Every queue is written ascending in the distance.
        auto write_layer = [&](const std::vector<int32_t> &members, int cap, bool with_keys, const std::string &coll) {
            const int64_t m = members.empty() ? n : (int64_t)members.size();
            std::vector<float> sub;
            const float *X = data_.data();
            if (!members.empty()) {
                sub.resize((size_t)m * (size_t)d);
                for (int64_t t = 0; t < m; t++) std::memcpy(sub.data() + (size_t)t * (size_t)d, Row((size_t)members[(size_t)t]), (size_t)d * sizeof(float));
                X = sub.data();
            }
            const int cap_r = m > (int64_t)cap + 1 ? cap / 8 : 0;  // reverse slots (a layer smaller than a queue links everybody anyway)
            const int cap_f = cap - cap_r;
            const int k = (int)std::min<int64_t>((int64_t)cap_f + 1, m);  // + 1: the vector itself may be among its own nearest
            const int64_t block = 65536;
            std::vector<int32_t> idx((size_t)std::min(block, m) * (size_t)k), cnt((size_t)std::min(block, m));
            std::vector<float> dist(idx.size());
            // forward lists of the whole layer: fwd_i / fwd_d, cap_f slots per member, flen of them used
            std::vector<int32_t> fwd_i((size_t)m * (size_t)cap_f), flen((size_t)m, 0), indeg((size_t)m, 0);
            std::vector<float> fwd_d((size_t)m * (size_t)cap_f);
            searcher_->invalidate(coll);
            for (int64_t q0 = 0; q0 < m; q0 += block) {
                const int64_t nq = std::min(block, m - q0);
                searcher_->search(coll, X, m, d, GORSE_METRIC_NEG_DOT, X + (size_t)q0 * (size_t)d, nq, k, idx.data(), dist.data(), cnt.data());
                for (int64_t t = 0; t < nq; t++) {
                    const int64_t self = q0 + t;
                    int32_t len = 0;
                    for (int32_t j = 0; j < cnt[(size_t)t] && len < cap_f; j++) {
                        const int32_t r = idx[(size_t)t * (size_t)k + (size_t)j];
                        if (r == (int32_t)self) continue;
                        fwd_i[(size_t)self * (size_t)cap_f + (size_t)len] = r;
                        fwd_d[(size_t)self * (size_t)cap_f + (size_t)len] = dist[(size_t)t * (size_t)k + (size_t)j];
                        indeg[(size_t)r]++;
                        len++;
                    }
                    flen[(size_t)self] = len;
                }
            }
            searcher_->invalidate(coll);
            // reverse candidates by target (CSR): r <- t for every forward link t -> r
            std::vector<int64_t> rptr((size_t)m + 1, 0);
            std::vector<int32_t> rsrc;
            std::vector<float> rdst;
            if (cap_r > 0) {
                for (int64_t t = 0; t < m; t++)
                    for (int32_t j = 0; j < flen[(size_t)t]; j++) rptr[(size_t)fwd_i[(size_t)t * (size_t)cap_f + (size_t)j] + 1]++;
                for (int64_t t = 0; t < m; t++) rptr[(size_t)t + 1] += rptr[(size_t)t];
                rsrc.resize((size_t)rptr[(size_t)m]);
                rdst.resize(rsrc.size());
                std::vector<int64_t> at(rptr.begin(), rptr.end() - 1);
                for (int64_t t = 0; t < m; t++)
                    for (int32_t j = 0; j < flen[(size_t)t]; j++) {
                        const int32_t r = fwd_i[(size_t)t * (size_t)cap_f + (size_t)j];
                        rsrc[(size_t)at[(size_t)r]] = (int32_t)t;
                        rdst[(size_t)at[(size_t)r]++] = fwd_d[(size_t)t * (size_t)cap_f + (size_t)j];  // -dot is symmetric, bit for bit
                    }
            }
            std::vector<std::pair<float, int32_t>> q;     // one queue: (distance, member)
            std::vector<std::pair<int32_t, int64_t>> cand;  // (incoming links of the source, position in rsrc)
            for (int64_t r = 0; r < m; r++) {
                q.clear();
                for (int32_t j = 0; j < flen[(size_t)r]; j++)
                    q.emplace_back(fwd_d[(size_t)r * (size_t)cap_f + (size_t)j], fwd_i[(size_t)r * (size_t)cap_f + (size_t)j]);
                if (cap_r > 0) {
                    cand.clear();
                    const int32_t *f0 = fwd_i.data() + (size_t)r * (size_t)cap_f;
                    for (int64_t e = rptr[(size_t)r]; e < rptr[(size_t)r + 1]; e++) {
                        const int32_t t = rsrc[(size_t)e];
                        if (std::find(f0, f0 + flen[(size_t)r], t) == f0 + flen[(size_t)r]) cand.emplace_back(indeg[(size_t)t], e);
                    }
                    const size_t take = std::min<size_t>((size_t)cap_r, cand.size());
                    std::partial_sort(cand.begin(), cand.begin() + (std::ptrdiff_t)take, cand.end(), [&](const auto &a, const auto &b) {
                        if (a.first != b.first) return a.first < b.first;
                        if (rdst[(size_t)a.second] != rdst[(size_t)b.second]) return rdst[(size_t)a.second] < rdst[(size_t)b.second];
                        return rsrc[(size_t)a.second] < rsrc[(size_t)b.second];
                    });
                    for (size_t c = 0; c < take; c++) {
                        q.emplace_back(rdst[(size_t)cand[c].second], rsrc[(size_t)cand[c].second]);
                        indeg[(size_t)rsrc[(size_t)cand[c].second]]++;
                    }
                    std::sort(q.begin(), q.end());
                }
                if (with_keys) put<int32_t>(w, members.empty() ? (int32_t)r : members[(size_t)r]);
                put<uint8_t>(w, 0);  // desc = false
                put<int32_t>(w, (int32_t)q.size());
                for (const auto &e : q) {
                    put<int32_t>(w, members.empty() ? e.second : members[(size_t)e.second]);
                    put<float>(w, e.first);
                }
            }
        };
        if (n > 0) write_layer({}, kM0, false, std::string(kCollection) + "#hnsw0");
        put<int64_t>(w, (int64_t)top);  // len(upperNeighbors)
        int32_t enter = 0;
        for (int L = 1; L <= top; L++) {
            std::vector<int32_t> members;
            for (int64_t i = 0; i < n; i++)
                if (level[(size_t)i] >= L) members.push_back((int32_t)i);
            put<int32_t>(w, (int32_t)members.size());
            write_layer(members, kM, true, std::string(kCollection) + "#hnsw" + std::to_string(L));
            if (L == top) enter = members.front();
        }
        put<int32_t>(w, enter);
        put_gob(w, gob::encode_int(n));
        for (const auto &id : items_) put_gob(w, gob::encode_string(id));
        return w;
    }
    void Unmarshal(const std::string &blob) {
        size_t at = 0;
        items_.clear();
        data_.clear();
        if (searcher_) searcher_->invalidate(kCollection);
        timestamp_ = gob::decode_time_unix_nanos(bytes(blob, at, get<int32_t>(blob, at)));
        dimension_ = (int)gob::decode_int(bytes(blob, at, get<int32_t>(blob, at)));
        int64_t n;
        if (blob.size() - at >= 4 && blob.compare(at, 4, "GHIP") == 0) {  // this library's index section
            at += 4;
            const int64_t version = get<int64_t>(blob, at);
            if (version != 1) throw std::runtime_error("MatrixFactorizationItems index section version " + std::to_string(version) + " is not supported");
            at += 3 * 8 + 8 + 4;
            if (at > blob.size()) throw std::runtime_error("unexpected EOF");
            n = get<int64_t>(blob, at);
            const int32_t d = get<int32_t>(blob, at);
            if (n < 0 || d != dimension_ || (n > 0 && (blob.size() - at) / sizeof(float) / (size_t)std::max(d, 1) < (size_t)n))
                throw std::runtime_error("unexpected EOF");
            data_.resize((size_t)n * (size_t)d);
            std::memcpy(data_.data(), blob.data() + at, data_.size() * sizeof(float));
            at += data_.size() * sizeof(float);
        } else {  // the reference's: HNSW.Marshal (hnsw.go:278-337)
            at += 4 + 4 * 8;  // levelFactor, maxConnection, maxConnection0, ef, efConstruction
            if (at > blob.size()) throw std::runtime_error("unexpected EOF");
            n = get<int64_t>(blob, at);
            if (n < 0) throw std::runtime_error("negative vector count");
            for (int64_t i = 0; i < n; i++) {
                const std::vector<float> v = gob::decode_f32_slice(bytes(blob, at, get<int32_t>(blob, at)));
                if ((int)v.size() != dimension_)
                    throw std::runtime_error("vector " + std::to_string(i) + " has " + std::to_string(v.size()) + " dimensions");
                data_.insert(data_.end(), v.begin(), v.end());
            }
            for (int64_t i = 0; i < n; i++) skip_queue(blob, at);  // bottom layer
            const int64_t layers = get<int64_t>(blob, at);
            for (int64_t l = 0; l < layers; l++) {
                const int32_t m = get<int32_t>(blob, at);
                for (int32_t e = 0; e < m; e++) {
                    (void)get<int32_t>(blob, at);
                    skip_queue(blob, at);
                }
            }
            (void)get<int32_t>(blob, at);  // enterPoint
        }
        const int64_t numItems = gob::decode_int(bytes(blob, at, get<int32_t>(blob, at)));
        if (numItems != n) throw std::runtime_error("ids and vectors differ in number");
        for (int64_t i = 0; i < numItems; i++) items_.push_back(gob::decode_string(bytes(blob, at, get<int32_t>(blob, at))));
    }

private:
    template <typename T>
    static void put(std::string &w, T v) { w.append((const char *)&v, sizeof(T)); }
    static void put_gob(std::string &w, const std::string &stream) {  // encoding.WriteGob: int32 byte count + the stream
        put<int32_t>(w, (int32_t)stream.size());
        w += stream;
    }
    template <typename T>
    static T get(const std::string &b, size_t &at) {
        if (b.size() < at || b.size() - at < sizeof(T)) throw std::runtime_error("unexpected EOF");
        T v;
        std::memcpy(&v, b.data() + at, sizeof(T));
        at += sizeof(T);
        return v;
    }
    static std::string bytes(const std::string &b, size_t &at, int32_t n) {
        if (n < 0 || b.size() < at || b.size() - at < (size_t)n) throw std::runtime_error("unexpected EOF");
        std::string out = b.substr(at, (size_t)n);
        at += (size_t)n;
        return out;
    }
    static void skip_queue(const std::string &b, size_t &at) {  // PriorityQueue.Marshal, pq.go:128-133
        (void)get<uint8_t>(b, at);
        const int32_t len = get<int32_t>(b, at);
        if (len < 0 || b.size() - at < (size_t)len * 8) throw std::runtime_error("unexpected EOF");
        at += (size_t)len * 8;
    }
    static constexpr const char *kCollection = "matrix_factorization_items";
    int64_t timestamp_ = 0;
    int dimension_ = 0;
    std::vector<std::string> items_;
    std::vector<float> data_;
    std::shared_ptr<vectors::Searcher> searcher_;
};

// The publishing step of the master after a fit (master/tasks.go:925-969): the predictable items' factors go into a fresh
// Dot collection collaborative_filtering_<modelId> (batches of batchSize, Id / IsHidden / Categories from the item table,
// Timestamp = the model id), the predictable users' factors into the MatrixFactorizationUsers blob the workers download.
// Model = anything with the cf.MatrixFactorization accessors (cf::BPR / cf::ALS); hidden[i] / categories[i] belong to item
// index i (data.Item.IsHidden / .Categories; either vector may be shorter than the item count = defaults).
template <typename Model>
MatrixFactorizationUsers PublishCollaborativeFiltering(Model &model, vectors::HipDatabase &db, int64_t modelId,
                                                       const std::vector<bool> &hidden,
                                                       const std::vector<std::vector<std::string>> &categories, int batchSize = 1024) {
    const std::string collection = "collaborative_filtering_" + std::to_string(modelId);  // database.go:52-54
    const int d = model.NFactors();
    const int32_t nItems = model.GetItemIndex()->Count(), nUsers = model.GetUserIndex()->Count();
    db.AddCollection(collection, d, vectors::Dot);
    for (int32_t start = 0; start < nItems; start += batchSize) {
        std::vector<vectors::Vector> batch;
        for (int32_t i = start; i < std::min<int32_t>(start + batchSize, nItems); i++) {
            if (!model.IsItemPredictable(i)) continue;
            vectors::Vector v;
            model.GetItemIndex()->String(i, v.Id);
            v.Values.assign(model.GetItemFactor(i), model.GetItemFactor(i) + d);
            v.IsHidden = (size_t)i < hidden.size() && hidden[(size_t)i];
            if ((size_t)i < categories.size()) v.Categories = categories[(size_t)i];
            v.TimestampMs = modelId;
            batch.push_back(std::move(v));
        }
        if (!batch.empty()) db.AddVectors(collection, batch);
    }
    MatrixFactorizationUsers users;
    for (int32_t u = 0; u < nUsers; u++) {
        std::string id;
        if (model.GetUserIndex()->String(u, id) && model.IsUserPredictable(u))
            users.Add(id, std::vector<float>(model.GetUserFactor(u), model.GetUserFactor(u) + d));
    }
    return users;
}

}  // namespace logics
}  // namespace gorse
