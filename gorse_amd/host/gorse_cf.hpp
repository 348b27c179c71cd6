// gorse_cf.hpp -- host side of the CF hot path, above the C ABI (include/gorse_hip.h).
//
// The reference's host code for this path is Go (model/cf, dataset, common/heap, common/ann); the Go
// toolchain is absent in the build image, so the same interfaces are mirrored here in C++ -- same
// names, argument meaning and error behaviour -- and everything numeric goes through the C ABI
// exactly as the cgo files of INTEGRATION.md would.  Nothing here computes factors, scores or
// top-k on the CPU: no GPU, no result.
//
//   dataset::FreqDict, dataset::Dataset (CFSplit)      dataset/dict.go, dataset/dataset.go:40-253
//   model::Params                                       model/params.go
//   cf::FitConfig, cf::Score, cf::MatrixFactorization   model/cf/model.go:44-127
//   cf::BPR, cf::ALS  (Fit / Predict / Marshal ...)     model/cf/model.go:367-792
//   cf::Evaluate, cf::NDCG ...                          model/cf/evaluator.go
//   heap::TopKFilter, heap::PriorityQueue               common/heap/filter.go, pq.go
//   ann::Index, ann::Bruteforce                         common/ann/ann.go, bruteforce.go
#pragma once
#include <climits>
#include <cmath>
#include <cstdint>
#include <functional>
#include <istream>
#include <map>
#include <memory>
#include <ostream>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/gorse_hip.h"
#include "../csrc/goheap.hpp"
#include "tpe.hpp"

namespace gorse {

struct HipError : std::runtime_error {
    int32_t code;
    HipError(int32_t c, const std::string &m) : std::runtime_error(m), code(c) {}
};
inline void check(int32_t rc) {
    if (rc != GORSE_OK) throw HipError(rc, gorse_hip_last_error());
}

// ------------------------------------------------------------------------------------------
namespace util {
// util.RandomGenerator (common/util/random.go:25-132).  The reference wraps Go's math/rand, whose
// stream cannot be reproduced outside Go; this one keeps the call surface and the
// distributions on a Philox4x32-10 stream (same generator the device sampler uses).
class RandomGenerator {
   public:
    explicit RandomGenerator(int64_t seed = 0) { g_.init((uint64_t)seed, 0, 0); }
    // a generator on one of the library's counter-based streams (csrc/common.hpp Philox: seed, epoch word, sample word)
    RandomGenerator(int64_t seed, uint64_t epoch, uint64_t sample) { g_.init((uint64_t)seed, epoch, sample); }
    int32_t Int31n(int32_t n) { return g_.int31n(n); }
    int Intn(int n) { return (int)g_.int31n((int32_t)n); }
    int64_t Int63() { return ((int64_t)g_.int31() << 32) | ((int64_t)g_.int31() << 1) | (g_.int31() & 1); }
    double Float64() { return (double)(Int63() >> 10) / (double)(1ll << 53); }
    double NormFloat64() {  // Box-Muller (Go uses a ziggurat; same distribution)
        if (have_) {
            have_ = false;
            return spare_;
        }
        double u1, u2;
        do u1 = Float64();
        while (u1 <= 0.0);
        u2 = Float64();
        double r = std::sqrt(-2.0 * std::log(u1)), t = 6.283185307179586476925286766559 * u2;
        spare_ = r * std::sin(t);
        have_ = true;
        return r * std::cos(t);
    }
    // NormalVector / NormalMatrix: float32(NormFloat64())*stdDev + mean, row-major draw order
    void NormalMatrix(int64_t row, int64_t col, float mean, float stdDev, std::vector<float> &out) {
        out.resize((size_t)(row * col));
        for (auto &x : out) x = (float)NormFloat64() * stdDev + mean;
    }
    // SampleInt32 (random.go:108-132)
    std::vector<int32_t> SampleInt32(int32_t low, int32_t high, int n, const std::set<int32_t> &exclude) {
        std::set<int32_t> ex(exclude);
        std::vector<int32_t> sampled;
        const int32_t len = high - low;
        if (n >= (int)len - (int)ex.size()) {
            for (int32_t i = low; i < high; i++)
                if (!ex.count(i)) {
                    sampled.push_back(i);
                    ex.insert(i);
                }
        } else {
            while ((int)sampled.size() < n) {
                int32_t v = Int31n(len) + low;
                if (!ex.count(v)) {
                    sampled.push_back(v);
                    ex.insert(v);
                }
            }
        }
        return sampled;
    }

   private:
    struct Px {  // host copy of csrc/common.hpp's Philox (kept header-only here)
        uint32_t c0, c1, c2, c3, k0, k1, buf[4];
        int pos;
        void init(uint64_t seed, uint64_t epoch, uint64_t sample) {
            c0 = (uint32_t)sample, c1 = (uint32_t)(sample >> 32), c2 = 0, c3 = (uint32_t)epoch;
            k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), pos = 4;
        }
        void block() {
            uint32_t a0 = c0, a1 = c1, a2 = c2, a3 = c3, x0 = k0, x1 = k1;
            for (int r = 0; r < 10; r++) {
                uint64_t p0 = (uint64_t)0xD2511F53u * a0, p1 = (uint64_t)0xCD9E8D57u * a2;
                uint32_t n0 = (uint32_t)(p1 >> 32) ^ a1 ^ x0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ a3 ^ x1,
                         n3 = (uint32_t)p0;
                a0 = n0, a1 = n1, a2 = n2, a3 = n3, x0 += 0x9E3779B9u, x1 += 0xBB67AE85u;
            }
            buf[0] = a0, buf[1] = a1, buf[2] = a2, buf[3] = a3, c2++, pos = 0;
        }
        uint32_t int31() {
            if (pos == 4) block();
            return buf[pos++] >> 1;
        }
        int32_t int31n(int32_t n) {
            if ((n & (n - 1)) == 0) return (int32_t)(int31() & (uint32_t)(n - 1));
            uint32_t mx = (uint32_t)((1u << 31) - 1 - (1u << 31) % (uint32_t)n), v = int31();
            while (v > mx) v = int31();
            return (int32_t)(v % (uint32_t)n);
        }
    } g_;
    bool have_ = false;
    double spare_ = 0;
};
}  // namespace util

// ------------------------------------------------------------------------------------------
namespace heap {
template <typename T>
struct Elem {
    T Value;
    float Weight;
    bool operator==(const Elem &o) const { return Value == o.Value && Weight == o.Weight; }
};

// heap.TopKFilter (filter.go:23-59) for int32 values: literal container/heap procedure.
class TopKFilter {
   public:
    explicit TopKFilter(int k) : k_(k), v_((size_t)k + 2), w_((size_t)k + 2), h_(v_.data(), w_.data()) {}
    void Push(int32_t item, float weight) {
        h_.push(item, weight);
        if (h_.n > k_) h_.pop();
    }
    int Len() const { return h_.n; }
    std::vector<int32_t> PopAllValues() {
        std::vector<int32_t> items((size_t)h_.n);
        for (int i = (int)items.size() - 1; i >= 0; i--) {
            h_.pop();
            items[(size_t)i] = h_.v[h_.n];
        }
        return items;
    }
    std::vector<Elem<int32_t>> PopAll() {
        std::vector<Elem<int32_t>> r((size_t)h_.n);
        for (int i = (int)r.size() - 1; i >= 0; i--) {
            h_.pop();
            r[(size_t)i] = {h_.v[h_.n], h_.w[h_.n]};
        }
        return r;
    }

   private:
    int k_;
    std::vector<int32_t> v_;
    std::vector<float> w_;
    GoHeap<false> h_;
};

// heap.PriorityQueue (pq.go:67-131): de-duplicating queue over container/heap.
class PriorityQueue {
   public:
    explicit PriorityQueue(bool desc) : desc_(desc) {}
    void Push(int32_t v, float weight) {
        if (std::isnan(weight)) throw std::invalid_argument("NaN weight is forbidden");
        if (lookup_.count(v)) return;
        v_.push_back(v);
        w_.push_back(weight);
        up((int)v_.size() - 1);
        lookup_.insert(v);
    }
    std::pair<int32_t, float> Pop() {
        int n = (int)v_.size() - 1;
        swap(0, n);
        down(0, n);
        auto r = std::make_pair(v_.back(), w_.back());
        v_.pop_back();
        w_.pop_back();
        return r;
    }
    std::pair<int32_t, float> Peek() const { return {v_[0], w_[0]}; }
    int Len() const { return (int)v_.size(); }
    std::vector<int32_t> Values() const { return v_; }
    PriorityQueue Clone() const {
        PriorityQueue p(desc_);
        p.v_ = v_;
        p.w_ = w_;
        return p;
    }
    PriorityQueue Reverse() const {
        PriorityQueue p(!desc_);
        for (size_t i = 0; i < v_.size(); i++) p.Push(v_[i], w_[i]);
        return p;
    }

   private:
    bool less(int i, int j) const { return desc_ ? w_[i] > w_[j] : w_[i] < w_[j]; }
    void swap(int i, int j) {
        std::swap(v_[i], v_[j]);
        std::swap(w_[i], w_[j]);
    }
    void up(int j) {
        for (;;) {
            int i = (j - 1) / 2;
            if (i == j || !less(j, i)) break;
            swap(i, j);
            j = i;
        }
    }
    void down(int i0, int n) {
        int i = i0;
        for (;;) {
            int j1 = 2 * i + 1;
            if (j1 >= n || j1 < 0) break;
            int j = j1, j2 = j1 + 1;
            if (j2 < n && less(j2, j1)) j = j2;
            if (!less(j, i)) break;
            swap(i, j);
            i = j;
        }
    }
    bool desc_;
    std::vector<int32_t> v_;
    std::vector<float> w_;
    std::set<int32_t> lookup_;
};
}  // namespace heap

// ------------------------------------------------------------------------------------------
namespace dataset {
// dataset.FreqDict (dict.go)
class FreqDict {
   public:
    int32_t Count() const { return (int32_t)is_.size(); }
    int32_t Add(const std::string &s) {
        auto it = si_.find(s);
        if (it != si_.end()) {
            cnt_[(size_t)it->second]++;
            return it->second;
        }
        int32_t y = (int32_t)is_.size();
        si_[s] = y;
        is_.push_back(s);
        cnt_.push_back(1);
        return y;
    }
    int32_t AddNoCount(const std::string &s) {
        auto it = si_.find(s);
        if (it != si_.end()) return it->second;
        int32_t y = (int32_t)is_.size();
        si_[s] = y;
        is_.push_back(s);
        cnt_.push_back(0);
        return y;
    }
    int32_t Id(const std::string &s) const {
        auto it = si_.find(s);
        return it == si_.end() ? -1 : it->second;
    }
    bool String(int32_t id, std::string &out) const {
        if (id < 0 || id >= (int32_t)is_.size()) return false;
        out = is_[(size_t)id];
        return true;
    }
    int32_t Freq(int32_t id) const { return id >= 0 && id < (int32_t)cnt_.size() ? cnt_[(size_t)id] : 0; }

   private:
    std::unordered_map<std::string, int32_t> si_;
    std::vector<std::string> is_;
    std::vector<int32_t> cnt_;
};

// dataset.Dataset restricted to its CFSplit surface (dataset.go:40-59, 78-253).
class Dataset {
   public:
    Dataset() : userDict_(std::make_shared<FreqDict>()), itemDict_(std::make_shared<FreqDict>()) {}
    // a split sharing the dictionaries of another dataset (LoadDataFromBuiltIn, dataset.go:410-414)
    explicit Dataset(const Dataset &shareDicts, bool) : userDict_(shareDicts.userDict_), itemDict_(shareDicts.itemDict_) {
        userFeedback_.resize(shareDicts.userFeedback_.size());
        itemFeedback_.resize(shareDicts.itemFeedback_.size());
    }
    void AddUser(const std::string &userId) {
        userDict_->AddNoCount(userId);
        if (userFeedback_.size() < (size_t)userDict_->Count()) userFeedback_.resize((size_t)userDict_->Count());
    }
    void AddItem(const std::string &itemId) {
        itemDict_->AddNoCount(itemId);
        if (itemFeedback_.size() < (size_t)itemDict_->Count()) itemFeedback_.resize((size_t)itemDict_->Count());
    }
    // AddFeedback (dataset.go:231-240)
    void AddFeedback(const std::string &userId, const std::string &itemId) {
        int32_t u = userDict_->Add(userId), i = itemDict_->Add(itemId);
        if (userFeedback_.size() <= (size_t)u) userFeedback_.resize((size_t)u + 1);
        if (itemFeedback_.size() <= (size_t)i) itemFeedback_.resize((size_t)i + 1);
        itemFeedback_[(size_t)i].push_back(u);
        userFeedback_[(size_t)u].push_back(i);
        numFeedback_++;
    }
    // bulk form used by bindings: dense indices == ids (loadTrain creates users/items 0..max, dataset.go:441-451)
    void AddFeedbackIndexed(int32_t u, int32_t i) {
        while (userDict_->Count() <= u) AddUser(std::to_string(userDict_->Count()));
        while (itemDict_->Count() <= i) AddItem(std::to_string(itemDict_->Count()));
        AddFeedback(std::to_string(u), std::to_string(i));
    }
    int CountUsers() const { return (int)std::max<size_t>(userFeedback_.size(), (size_t)userDict_->Count()); }
    int CountItems() const { return (int)std::max<size_t>(itemFeedback_.size(), (size_t)itemDict_->Count()); }
    int CountFeedback() const { return numFeedback_; }
    std::shared_ptr<FreqDict> GetUserDict() const { return userDict_; }
    std::shared_ptr<FreqDict> GetItemDict() const { return itemDict_; }
    const std::vector<std::vector<int32_t>> &GetUserFeedback() const {
        const_cast<Dataset *>(this)->userFeedback_.resize((size_t)CountUsers());
        return userFeedback_;
    }
    const std::vector<std::vector<int32_t>> &GetItemFeedback() const {
        const_cast<Dataset *>(this)->itemFeedback_.resize((size_t)CountItems());
        return itemFeedback_;
    }
    // SplitCF (dataset.go:258-318): leave-one-out per user.  numTestUsers <= 0 or >= CountUsers(): every user with feedback
    // gives one (rng.Intn) of its items to the test set; otherwise only numTestUsers sampled users do.  The splits share
    // this dataset's dictionaries.  (Which item / which users depends on the generator: Go's stream is not reproducible
    // here, SURVEY.md 8c -- the structure and the counts are.)
    std::pair<Dataset, Dataset> SplitCF(int numTestUsers, int64_t seed) const {
        Dataset train(*this, true), test(*this, true);
        const int U = CountUsers();
        train.userFeedback_.resize((size_t)U), test.userFeedback_.resize((size_t)U);
        train.itemFeedback_.resize((size_t)CountItems()), test.itemFeedback_.resize((size_t)CountItems());
        util::RandomGenerator rng(seed);
        const auto &uf = GetUserFeedback();
        auto leave_one_out = [&](int32_t u) {
            const auto &row = uf[(size_t)u];
            if (row.empty()) return;
            const int k = rng.Intn((int)row.size());
            test.add_indexed(u, row[(size_t)k]);
            for (size_t i = 0; i < row.size(); i++)
                if ((int)i != k) train.add_indexed(u, row[i]);
        };
        if (numTestUsers >= U || numTestUsers <= 0) {
            for (int32_t u = 0; u < U; u++) leave_one_out(u);
        } else {
            const std::vector<int32_t> testUsers = rng.SampleInt32(0, U, numTestUsers, {});
            for (int32_t u : testUsers) leave_one_out(u);
            const std::set<int32_t> chosen(testUsers.begin(), testUsers.end());
            for (int32_t u = 0; u < U; u++)
                if (!chosen.count(u))
                    for (int32_t i : uf[(size_t)u]) train.add_indexed(u, i);
        }
        return {std::move(train), std::move(test)};
    }
    // LoadDataFromBuiltIn without the download (dataset.go:398-490): train.txt = "user<TAB>item[<TAB>...]"
    // per line, users / items 0..max all created; test.txt = "(user,item)<TAB>neg<TAB>neg..." per line, the negatives kept
    // per user and -- like the reference, which looks them up with itemDict.Add -- counted in the item dictionary.
    static std::pair<Dataset, Dataset> LoadNCF(std::istream &trainFile, std::istream &testFile) {
        Dataset train;
        std::string line;
        auto split = [](const std::string &l, char sep) {
            std::vector<std::string> out(1);
            for (char c : l)
                if (c == sep)
                    out.emplace_back();
                else
                    out.back().push_back(c);
            return out;
        };
        auto parse = [](const std::string &f) {  // util.ParseInt[int32]
            size_t used = 0;
            long long v = 0;
            try {
                v = std::stoll(f, &used);
            } catch (const std::exception &) {
                used = 0;
            }
            if (f.empty() || used != f.size() || v < INT32_MIN || v > INT32_MAX) throw std::invalid_argument("invalid integer: " + f);
            return (int32_t)v;
        };
        while (std::getline(trainFile, line)) {
            const auto f = split(line, '\t');
            if (f.size() < 2) throw std::invalid_argument("wrong format: " + line);
            const int32_t u = parse(f[0]), i = parse(f[1]);
            for (int32_t t = train.userDict_->Count(); t <= u; t++) train.AddUser(std::to_string(t));
            for (int32_t t = train.itemDict_->Count(); t <= i; t++) train.AddItem(std::to_string(t));
            train.AddFeedback(f[0], f[1]);
        }
        Dataset test(train, true);
        test.negatives_.resize(train.userFeedback_.size());
        while (std::getline(testFile, line)) {
            const auto f = split(line, '\t');
            const std::string &pos = f[0];
            if (pos.size() < 2 || pos.front() != '(' || pos.back() != ')') throw std::invalid_argument("wrong format: " + line);
            const auto pair = split(pos.substr(1, pos.size() - 2), ',');
            if (pair.size() < 2) throw std::invalid_argument("wrong format: " + line);
            test.AddFeedback(pair[0], pair[1]);
            const int32_t u = parse(pair[0]);
            if (u < 0 || (size_t)u >= test.negatives_.size()) throw std::out_of_range("index out of range: " + pair[0]);  // a Go panic
            std::vector<int32_t> negs;
            for (size_t t = 1; t < f.size(); t++) negs.push_back(test.itemDict_->Add(f[t]));
            test.negatives_[(size_t)u] = std::move(negs);
        }
        return {std::move(train), std::move(test)};
    }
    const std::vector<std::vector<int32_t>> &Negatives() const { return negatives_; }
    // GetUserIDF / GetItemIDF (dataset.go:160-180): idf = math32.Log(1 + float32(#other side) / float32(freq)); the
    // weights of the sparse "users" item-to-item and "items" user-to-user vectors (logics/vector_writer.go:192-209).
    // math32.Log (chewxy/math32 v1.11.1) restated as the float64 logarithm narrowed to float32.
    std::vector<float> GetUserIDF() const { return idf(*userDict_, CountItems()); }
    std::vector<float> GetItemIDF() const { return idf(*itemDict_, CountUsers()); }
    void SetNegatives(int32_t user, std::vector<int32_t> negs) {
        if (negatives_.size() < (size_t)CountUsers()) negatives_.resize((size_t)CountUsers());
        negatives_[(size_t)user] = std::move(negs);
    }
    // SampleUserNegatives (dataset.go:242-253): cached.  The reference walks the users with ONE generator seeded 0; its
    // math/rand stream is not reproducible, so user u draws from its own stream (0, "neg", u) instead --
    // the stream gorse_mf_sample_user_negatives uses on the device, where all users are sampled at once: the host loop
    // below and the device produce the same lists.
    static constexpr uint64_t kNegStream = 0x6e6567ull;
    bool HasNegatives() const {
        for (auto &n : negatives_)
            if (!n.empty()) return true;
        return false;
    }
    const std::vector<std::vector<int32_t>> &SampleUserNegatives(const Dataset &excludeSet, int numCandidates) {
        if (!HasNegatives()) {
            negatives_.assign((size_t)CountUsers(), {});
            const auto &mine = GetUserFeedback();
            const auto &other = excludeSet.GetUserFeedback();
            for (int u = 0; u < CountUsers(); u++) {
                std::set<int32_t> ex;
                if ((size_t)u < mine.size()) ex.insert(mine[(size_t)u].begin(), mine[(size_t)u].end());
                if ((size_t)u < other.size()) ex.insert(other[(size_t)u].begin(), other[(size_t)u].end());
                util::RandomGenerator rng(0, kNegStream, (uint64_t)u);
                negatives_[(size_t)u] = rng.SampleInt32(0, (int32_t)CountItems(), numCandidates, ex);
            }
        }
        if (negatives_.size() < (size_t)CountUsers()) negatives_.resize((size_t)CountUsers());
        return negatives_;
    }
    // the device's lists (U x n padded with -1, lengths) become this split's cached negatives
    void SetSampledNegatives(const std::vector<int32_t> &neg, const std::vector<int32_t> &len, int n) {
        negatives_.assign((size_t)CountUsers(), {});
        for (size_t u = 0; u < negatives_.size() && u < len.size(); u++)
            negatives_[u].assign(neg.begin() + (ptrdiff_t)(u * (size_t)n), neg.begin() + (ptrdiff_t)(u * (size_t)n) + len[u]);
    }

   private:
    void add_indexed(int32_t u, int32_t i) {  // a feedback between existing dense indices, dictionaries untouched
        userFeedback_[(size_t)u].push_back(i);
        itemFeedback_[(size_t)i].push_back(u);
        numFeedback_++;
    }
    static std::vector<float> idf(const FreqDict &dict, int other) {
        std::vector<float> out((size_t)dict.Count());
        for (int32_t t = 0; t < dict.Count(); t++)
            out[(size_t)t] = (float)std::log((double)(1.0f + (float)other / (float)dict.Freq(t)));
        return out;
    }
    std::shared_ptr<FreqDict> userDict_, itemDict_;
    std::vector<std::vector<int32_t>> userFeedback_, itemFeedback_, negatives_;
    int numFeedback_ = 0;
};
}  // namespace dataset

// ------------------------------------------------------------------------------------------
namespace model {
// model.Params (params.go): names as in the reference.
inline const char *Lr = "Lr", *Reg = "Reg", *NEpochs = "NEpochs", *NFactors = "NFactors", *RandomState = "RandomState",
                  *InitMean = "InitMean", *InitStdDev = "InitStdDev", *Alpha = "Alpha";
class Params : public std::map<std::string, double> {
   public:
    using std::map<std::string, double>::map;
    int GetInt(const std::string &n, int d) const {
        auto it = find(n);
        return it == end() ? d : (int)it->second;
    }
    int64_t GetInt64(const std::string &n, int64_t d) const {
        auto it = find(n);
        return it == end() ? d : (int64_t)it->second;
    }
    float GetFloat32(const std::string &n, float d) const {
        auto it = find(n);
        return it == end() ? d : (float)it->second;
    }
};
}  // namespace model

// ------------------------------------------------------------------------------------------
namespace cf {
struct Score {
    float NDCG = 0, Precision = 0, Recall = 0;
};
// One gorse_mf handle (dataset CSR on the device + factor matrices of a fixed nFactors) lent to successive models that fit
// the same training set: Fit then only uploads fresh factors (gorse_mf_set_factors) instead of the whole dataset.
struct ResidentDataset {
    gorse_mf *h = nullptr;
    const void *trainSet = nullptr;  // identity of the dataset::Dataset the handle was built from
    int64_t U = 0, I = 0, N = 0;
    int nFactors = 0, device = 0;
    bool with_items = false;
    int uploads = 0, reuses = 0;  // how often the dataset crossed PCIe / was found resident
    ~ResidentDataset() {
        if (h) gorse_mf_destroy(h);
    }
};
// FitConfig (model.go:50-80) plus the two hooks a Go context / monitor span provide there.
struct FitConfig {
    int Jobs = 1, Verbose = 10, Candidates = 100, TopK = 10, Patience = 0;
    const volatile int32_t *Cancel = nullptr;        // ctx.Done()
    std::function<void(int)> OnEpoch;                 // span.Add(1): a progress count -- fires when the epoch has been ISSUED (the epochs between two
                                                      // evaluations are only enqueued; every epoch is done before an evaluation and before Fit returns)
    std::function<void(const std::string &)> Log;     // zap logger lines ("fit bpr e/E ...")
    int Device = 0;
    // a dataset kept on the device across Fit calls (the trials of a ModelSearch, the fit periods of the
    // master); null = every Fit uploads its own copy, as round 1 did
    struct ResidentDataset *Resident = nullptr;
    FitConfig &SetVerbose(int v) { Verbose = v; return *this; }
    FitConfig &SetJobs(int j) { Jobs = j; return *this; }
    FitConfig &SetPatience(int p) { Patience = p; return *this; }
};
inline FitConfig NewFitConfig() { return FitConfig(); }
// epochs a Fit keeps in flight between two evaluations (BPR::Fit, gorse_mf_epoch_throttle): two -- one running, one whose preparation
// runs under it -- so that a cancelled context is seen within two epochs; the reference checks per sample (model.go:449)
constexpr int kEnqueueDepth = 2;

using TargetSet = std::set<int32_t>;
using Metric = float (*)(const TargetSet &, const std::vector<int32_t> &);
// evaluator.go:75-160 (math32.Log2 -> log2f)
inline float NDCG(const TargetSet &t, const std::vector<int32_t> &r) {
    float idcg = 0, dcg = 0;
    for (size_t i = 0; i < t.size() && i < r.size(); i++) idcg += 1.0f / log2f((float)i + 2.0f);
    for (size_t i = 0; i < r.size(); i++)
        if (t.count(r[i])) dcg += 1.0f / log2f((float)i + 2.0f);
    return dcg / idcg;
}
inline float Precision(const TargetSet &t, const std::vector<int32_t> &r) {
    float hit = 0;
    for (auto x : r)
        if (t.count(x)) hit++;
    return hit / (float)r.size();
}
inline float Recall(const TargetSet &t, const std::vector<int32_t> &r) {
    int hit = 0;
    for (auto x : r)
        if (t.count(x)) hit++;
    return (float)hit / (float)t.size();
}
inline float HR(const TargetSet &t, const std::vector<int32_t> &r) {
    for (auto x : r)
        if (t.count(x)) return 1;
    return 0;
}
inline float MAP(const TargetSet &t, const std::vector<int32_t> &r) {
    float sum = 0;
    int hit = 0;
    for (size_t i = 0; i < r.size(); i++)
        if (t.count(r[i])) {
            hit++;
            sum += (float)hit / (float)(i + 1);
        }
    return sum / (float)t.size();
}
inline float MRR(const TargetSet &t, const std::vector<int32_t> &r) {
    for (size_t i = 0; i < r.size(); i++)
        if (t.count(r[i])) return 1.0f / (float)(i + 1);
    return 0;
}

// goptuna.Trial (github.com/c-bata/goptuna v0.9.0, go.mod) as far as SuggestParams / ModelSearch.Objective call it
// (model.go:397-405, 588-596; optimize.go:61-64; optimize_test.go:95-99)
struct Trial {
    virtual ~Trial() = default;
    virtual std::string SuggestCategorical(const std::string &name, const std::vector<std::string> &choices) = 0;
    virtual double SuggestLogFloat(const std::string &name, double low, double high) = 0;
    virtual double SuggestDiscreteFloat(const std::string &name, double low, double high, double q) = 0;
};

// cf.MatrixFactorization (model.go:82-127) with its state resident on one MI355X.
class MatrixFactorization {
   public:
    virtual ~MatrixFactorization() { release(); }
    virtual const char *Name() const = 0;
    virtual model::Params SuggestParams(Trial &trial) = 0;  // model.Model (model/model.go): the search space of a trial
    virtual void SetParams(const model::Params &p) {
        Params = p;
        randState_ = Params.GetInt64(model::RandomState, 0);
        rng_ = util::RandomGenerator(randState_);
    }
    const model::Params &GetParams() const { return Params; }
    virtual Score Fit(dataset::Dataset &trainSet, dataset::Dataset &valSet, const FitConfig &config) = 0;

    std::shared_ptr<dataset::FreqDict> GetUserIndex() const { return UserIndex; }
    std::shared_ptr<dataset::FreqDict> GetItemIndex() const { return ItemIndex; }
    bool IsUserPredictable(int32_t u) const {
        return UserIndex && u >= 0 && u < UserIndex->Count() && (size_t)u < UserPredictable.size() && UserPredictable[(size_t)u];
    }
    bool IsItemPredictable(int32_t i) const {
        return ItemIndex && i >= 0 && i < ItemIndex->Count() && (size_t)i < ItemPredictable.size() && ItemPredictable[(size_t)i];
    }
    const float *GetUserFactor(int32_t u) const { return UserFactor.data() + (size_t)u * (size_t)nFactors_; }
    const float *GetItemFactor(int32_t i) const { return ItemFactor.data() + (size_t)i * (size_t)nFactors_; }
    int NFactors() const { return nFactors_; }
    int EpochsDone() const { return epochs_done_; }  // epochs the last Fit ran (early stopping / cancellation: fewer than NEpochs)
    // Predict (model.go:182-193): unknown ids -> 0 (+ warning)
    float Predict(const std::string &userId, const std::string &itemId) {
        int32_t u = UserIndex ? UserIndex->Id(userId) : -1, i = ItemIndex ? ItemIndex->Id(itemId) : -1;
        return internalPredict(u, i);
    }
    // internalPredict (model.go:195-203) = floats.Dot on the device, in the reference's AVX512 order
    float internalPredict(int32_t u, int32_t i) {
        float r = 0;
        if (u >= 0 && i >= 0) {
            ensure_resident();
            check(gorse_mf_score(h_, &u, &i, 1, &r));
        }
        return r;
    }
    std::vector<float> internalPredictMany(const std::vector<int32_t> &us, const std::vector<int32_t> &is) {
        if (us.size() != is.size()) throw std::invalid_argument("floats: slice lengths do not match");
        std::vector<float> out(us.size());
        if (!us.empty()) {
            ensure_resident();
            check(gorse_mf_score(h_, us.data(), is.data(), (int64_t)us.size(), out.data()));
        }
        return out;
    }
    void Clear() {
        UserIndex.reset();
        ItemIndex.reset();
        UserFactor.clear();
        ItemFactor.clear();
        release();
    }
    bool Invalid() const { return !UserIndex || !ItemIndex || ItemFactor.empty() || UserFactor.empty(); }

    // Marshal / Unmarshal (model.go:206-292): the Params block is a gob stream of map[ParamName]any (gob.hpp: restated
    // from the encoding/gob documentation, unpinned against a Go encoder), the counts are little-endian int64, the
    // LatentFactor records are byte-identical protobuf (protocol/encoding.proto:27-30, pbutil.WriteDelimited).
    void Marshal(std::ostream &w) const;
    void Unmarshal(std::istream &r);

    // SampleUserNegatives on the device: possible when the resident handle holds `trainSet` (a Fit in
    // progress).  Fills testSet's cached negatives and leaves the candidate lists of the users with test feedback resident, so
    // that every Evaluate of this Fit ranks without an upload (RankResident).
    bool SampleNegativesOnDevice(dataset::Dataset &testSet, const dataset::Dataset &trainSet, int numCandidates) {
        if (!h_ || handle_train_ != (const void *)&trainSet || testSet.CountUsers() != trainSet.CountUsers()) return false;
        const size_t U = (size_t)testSet.CountUsers();
        const auto &tf = testSet.GetUserFeedback();
        std::vector<int64_t> tptr(U + 1, 0);
        for (size_t u = 0; u < U; u++) tptr[u + 1] = tptr[u] + (u < tf.size() ? (int64_t)tf[u].size() : 0);
        std::vector<int32_t> tidx((size_t)tptr[U] + 1, 0);
        for (size_t u = 0; u < U && u < tf.size(); u++) std::copy(tf[u].begin(), tf[u].end(), tidx.begin() + tptr[u]);
        std::vector<int32_t> neg(U * (size_t)numCandidates), len(U);
        check(gorse_mf_sample_user_negatives(h_, tptr.data(), tidx.data(), numCandidates, 0, neg.data(), len.data()));
        testSet.SetSampledNegatives(neg, len, numCandidates);
        resident_eval_ = (const void *)&testSet;
        resident_candidates_ = numCandidates;
        resident_feedback_ = (int64_t)testSet.CountFeedback();
        check(gorse_mf_resident_generation(h_, &resident_gen_));
        return true;
    }
    // The lists on the handle are this model's own sampling of THIS split: same object, same size, same candidate count, and no
    // other sampling on the (possibly lent) handle since -- the handle's generation number still reads what ours returned.
    bool HasResidentCandidates(const dataset::Dataset &testSet, int numCandidates) const {
        if (!h_ || resident_eval_ != (const void *)&testSet || resident_candidates_ != numCandidates ||
            resident_feedback_ != (int64_t)testSet.CountFeedback())
            return false;
        uint64_t now = 0;
        return gorse_mf_resident_generation(h_, &now) == GORSE_OK && now != 0 && now == resident_gen_;
    }
    std::vector<std::vector<int32_t>> RankResident(std::vector<int32_t> &users, int topN) {
        int64_t nu = 0, nc = 0;
        check(gorse_mf_resident_candidates(h_, &nu, &nc));
        users.resize((size_t)nu);
        std::vector<int32_t> rank((size_t)nu * (size_t)topN), len((size_t)nu);
        if (nu > 0) check(gorse_mf_rank_resident(h_, topN, users.data(), rank.data(), len.data()));
        std::vector<std::vector<int32_t>> out((size_t)nu);
        for (size_t t = 0; t < (size_t)nu; t++)
            out[t].assign(rank.begin() + (ptrdiff_t)(t * (size_t)topN), rank.begin() + (ptrdiff_t)(t * (size_t)topN) + len[t]);
        return out;
    }

    // Rank lists for Evaluate: user -> candidates, on the device (evaluator.go:162-169)
    std::vector<std::vector<int32_t>> RankMany(const std::vector<int32_t> &users,
                                               const std::vector<std::vector<int32_t>> &cands, int topN) {
        ensure_resident();
        std::vector<int64_t> ptr(users.size() + 1, 0);
        for (size_t t = 0; t < users.size(); t++) ptr[t + 1] = ptr[t] + (int64_t)cands[t].size();
        std::vector<int32_t> flat((size_t)ptr.back());
        for (size_t t = 0; t < users.size(); t++) std::copy(cands[t].begin(), cands[t].end(), flat.begin() + ptr[t]);
        std::vector<int32_t> rank(users.size() * (size_t)topN), len(users.size());
        check(gorse_mf_rank(h_, (int64_t)users.size(), users.data(), ptr.data(), flat.data(), topN, rank.data(), len.data()));
        std::vector<std::vector<int32_t>> out(users.size());
        for (size_t t = 0; t < users.size(); t++)
            out[t].assign(rank.begin() + (ptrdiff_t)(t * (size_t)topN), rank.begin() + (ptrdiff_t)(t * (size_t)topN) + len[t]);
        return out;
    }

    model::Params Params;
    std::shared_ptr<dataset::FreqDict> UserIndex, ItemIndex;
    std::vector<bool> UserPredictable, ItemPredictable;
    std::vector<float> UserFactor, ItemFactor;  // flat row-major (the reference's [][]float32 rows)

   protected:
    void Init(const dataset::Dataset &trainSet);  // BaseMatrixFactorization.Init, model.go:129-146
    void create_handle(const dataset::Dataset &trainSet, bool with_items, int device, ResidentDataset *res = nullptr);
    void ensure_resident();
    void pull_factors() { check(gorse_mf_get_factors(h_, UserFactor.data(), ItemFactor.data())); }
    void release() {
        if (h_ && !borrowed_) gorse_mf_destroy(h_);  // a borrowed handle stays with its ResidentDataset
        h_ = nullptr;
        borrowed_ = false;
        handle_train_ = resident_eval_ = nullptr;
        resident_gen_ = 0;
    }
    util::RandomGenerator &GetRandomGenerator() { return rng_; }
    // the shared evaluate / early-stopping epoch loop of BPR.Fit and ALS.Fit (model.go:432-440, 496-518)
    Score fit_loop(const char *tag, int nEpochs, dataset::Dataset &trainSet, dataset::Dataset &valSet, const FitConfig &config,
                   const std::function<int32_t(int)> &run_epoch);

    int nFactors_ = 16;
    int64_t randState_ = 0;
    util::RandomGenerator rng_{0};
    gorse_mf *h_ = nullptr;
    bool borrowed_ = false;
    int epochs_done_ = 0;  // fit_loop
    int device_ = 0;
    const void *handle_train_ = nullptr;   // the training set the resident handle was created from (a Fit in progress)
    const void *resident_eval_ = nullptr;  // the test split whose candidate lists are resident on h_
    uint64_t resident_gen_ = 0;            // ... as of this sampling (gorse_mf_resident_generation), with this many candidates
    int resident_candidates_ = 0;          //     per user and this many test feedbacks
    int64_t resident_feedback_ = -1;
};

// Evaluate (evaluator.go:35-72): candidates = test positives ++ negatives; device rank lists; the
// metric sums are float32 additions in user order (== the reference with nJobs = 1).
std::vector<float> Evaluate(MatrixFactorization &estimator, dataset::Dataset &testSet, dataset::Dataset &trainSet, int topK,
                            int numCandidates, int nJobs, const std::vector<Metric> &scorers);

class BPR : public MatrixFactorization {
   public:
    explicit BPR(const model::Params &p = {}) { SetParams(p); }
    const char *Name() const override { return "bpr"; }
    void SetParams(const model::Params &p) override {  // model.go:382-391
        MatrixFactorization::SetParams(p);
        nFactors_ = Params.GetInt(model::NFactors, 16);
        nEpochs = Params.GetInt(model::NEpochs, 100);
        lr = Params.GetFloat32(model::Lr, 0.05f);
        reg = Params.GetFloat32(model::Reg, 0.01f);
        initMean = Params.GetFloat32(model::InitMean, 0);
        initStdDev = Params.GetFloat32(model::InitStdDev, 0.001f);
    }
    Score Fit(dataset::Dataset &trainSet, dataset::Dataset &valSet, const FitConfig &config) override;
    model::Params SuggestParams(Trial &trial) override {  // model.go:397-405
        model::Params p;
        p[model::NFactors] = 16;
        p[model::Lr] = trial.SuggestLogFloat(model::Lr, 0.001, 0.1);
        p[model::Reg] = trial.SuggestLogFloat(model::Reg, 0.001, 0.1);
        p[model::InitMean] = 0;
        p[model::InitStdDev] = trial.SuggestLogFloat(model::InitStdDev, 0.001, 0.1);
        return p;
    }
    int nEpochs = 100;
    float lr = 0.05f, reg = 0.01f, initMean = 0, initStdDev = 0.001f;
};

class ALS : public MatrixFactorization {
   public:
    explicit ALS(const model::Params &p = {}) { SetParams(p); }
    const char *Name() const override { return "als"; }
    void SetParams(const model::Params &p) override {  // model.go:577-586
        MatrixFactorization::SetParams(p);
        nFactors_ = Params.GetInt(model::NFactors, 16);
        nEpochs = Params.GetInt(model::NEpochs, 50);
        initMean = Params.GetFloat32(model::InitMean, 0);
        initStdDev = Params.GetFloat32(model::InitStdDev, 0.1f);
        reg = Params.GetFloat32(model::Reg, 0.06f);
        weight = Params.GetFloat32(model::Alpha, 0.001f);
    }
    Score Fit(dataset::Dataset &trainSet, dataset::Dataset &valSet, const FitConfig &config) override;
    model::Params SuggestParams(Trial &trial) override {  // model.go:588-596
        model::Params p;
        p[model::NFactors] = 16;
        p[model::InitMean] = 0;
        p[model::InitStdDev] = trial.SuggestLogFloat(model::InitStdDev, 0.001, 0.1);
        p[model::Reg] = trial.SuggestLogFloat(model::Reg, 0.001, 0.1);
        p[model::Alpha] = trial.SuggestLogFloat(model::Alpha, 0.001, 0.1);
        return p;
    }
    int nEpochs = 50;
    float reg = 0.06f, initMean = 0, initStdDev = 0.1f, weight = 0.001f;
};

// ---- hyper-parameter search: ModelSearch (optimize.go:28-85) + the study that drives it (master/tasks.go:1268-1316) ----
using ModelCreator = std::function<std::unique_ptr<MatrixFactorization>()>;
struct SearchResult {  // meta.Model[Score]
    std::string Type;
    model::Params Params;
    Score Score_;
};
class ModelSearch {
   public:
    // trainSet / valSet may be null for models whose Fit ignores them (optimize_test.go's mock).  With keepResident the
    // search lends ONE device copy of the training set to all its trials (every SuggestParams fixes NFactors = 16).
    ModelSearch(std::map<std::string, ModelCreator> models, dataset::Dataset *trainSet, dataset::Dataset *valSet,
                FitConfig config, bool keepResident = true)
        : creators_(std::move(models)), train_(trainSet), val_(valSet), config_(std::move(config)) {
        for (auto &kv : creators_) types_.push_back(kv.first);  // maps.Keys: any order in Go, sorted here
        if (keepResident) config_.Resident = &resident_;
    }
    double Objective(Trial &trial) {  // optimize.go:61-81
        if (creators_.empty()) throw std::runtime_error("no model to search");
        const std::string type = trial.SuggestCategorical("Model", types_);
        auto m = creators_.at(type)();
        m->SetParams(m->SuggestParams(trial));
        static dataset::Dataset none;
        const Score score = m->Fit(train_ ? *train_ : none, val_ ? *val_ : none, config_);
        if (score.NDCG > result_.Score_.NDCG) {
            result_.Type = type;
            result_.Params = m->GetParams();
            result_.Score_ = score;
        }
        if (OnTrial) OnTrial();  // span.Add(1)
        return (double)score.NDCG;
    }
    const SearchResult &Result() const { return result_; }
    const ResidentDataset &Resident() const { return resident_; }
    std::function<void()> OnTrial;

   private:
    std::map<std::string, ModelCreator> creators_;
    std::vector<std::string> types_;
    dataset::Dataset *train_, *val_;
    FitConfig config_;
    ResidentDataset resident_;  // declared after config_: destroyed first, while no model borrows it any more
    SearchResult result_;
};

// A trial that draws every parameter independently: uniform over the choices, log-uniform, uniform over the grid
// low, low + q, ... <= high.  This is goptuna's random sampling, which its TPE sampler also uses for its start-up trials
// (TpeTrial below); which values Go's generator would draw is unpinned like every math/rand stream.
class RandomTrial : public Trial {
   public:
    explicit RandomTrial(util::RandomGenerator &rng) : rng_(rng) {}
    std::string SuggestCategorical(const std::string &name, const std::vector<std::string> &choices) override {
        if (choices.empty()) throw std::invalid_argument("no choices for " + name);
        return choices[(size_t)rng_.Intn((int)choices.size())];
    }
    double SuggestLogFloat(const std::string &name, double low, double high) override {
        if (!(low > 0) || high < low) throw std::invalid_argument("bad log range for " + name);
        const double v = std::exp(std::log(low) + rng_.Float64() * (std::log(high) - std::log(low)));
        return record(name, std::min(std::max(v, low), high));
    }
    double SuggestDiscreteFloat(const std::string &name, double low, double high, double q) override {
        if (!(q > 0) || high < low) throw std::invalid_argument("bad grid for " + name);
        const int steps = (int)std::floor((high - low) / q + 1e-9) + 1;
        return record(name, low + q * (double)rng_.Intn(steps));
    }
    std::map<std::string, double> Values;  // what this trial drew, by parameter name

   private:
    double record(const std::string &name, double v) {
        Values[name] = v;
        return v;
    }
    util::RandomGenerator &rng_;
};

// A trial of the TPE sampler (tpe.hpp): every parameter is suggested from the finished trials that have it -- at random for
// the first ten, then the best of 24 candidates drawn from the Parzen estimator of the better trials.
class TpeTrial : public Trial {
   public:
    TpeTrial(util::RandomGenerator &rng, const std::vector<tpe::Finished> &done) : rng_(rng), done_(done), random_(rng) {}
    std::string SuggestCategorical(const std::string &name, const std::vector<std::string> &choices) override {
        if (choices.empty()) throw std::invalid_argument("no choices for " + name);
        std::vector<double> below, above;
        tpe::split(done_, name, below, above);
        int c;
        if ((int)(below.size() + above.size()) < tpe::kStartupTrials)
            c = rng_.Intn((int)choices.size());
        else
            c = tpe::suggest_categorical(below, above, (int)choices.size(), rng_);
        Values[name] = (double)c;
        return choices[(size_t)c];
    }
    double SuggestLogFloat(const std::string &name, double low, double high) override {
        if (!(low > 0) || high < low) throw std::invalid_argument("bad log range for " + name);
        std::vector<double> below, above;
        tpe::split(done_, name, below, above);
        if ((int)(below.size() + above.size()) < tpe::kStartupTrials) return Values[name] = random_.SuggestLogFloat(name, low, high);
        for (double &x : below) x = std::log(x);
        for (double &x : above) x = std::log(x);
        const double v = std::exp(tpe::suggest_numerical(below, above, std::log(low), std::log(high), 0.0, rng_));
        return Values[name] = std::min(std::max(v, low), high);
    }
    double SuggestDiscreteFloat(const std::string &name, double low, double high, double q) override {
        if (!(q > 0) || high < low) throw std::invalid_argument("bad grid for " + name);
        std::vector<double> below, above;
        tpe::split(done_, name, below, above);
        if ((int)(below.size() + above.size()) < tpe::kStartupTrials) return Values[name] = random_.SuggestDiscreteFloat(name, low, high, q);
        const double v = tpe::suggest_numerical(below, above, low - 0.5 * q, high + 0.5 * q, q, rng_);
        return Values[name] = std::min(std::max(v, low), high);
    }
    std::map<std::string, double> Values;  // what this trial drew (a categorical parameter: the position of the choice)

   private:
    util::RandomGenerator &rng_;
    const std::vector<tpe::Finished> &done_;
    RandomTrial random_;
};

// goptuna.CreateStudy(direction = maximize, sampler) + study.Optimize(objective, nTrials): the TPE sampler of
// master/tasks.go:1297-1300 by default, independent random trials on request (goptuna's default sampler)
class Study {
   public:
    enum class Sampler { TPE, Random };
    explicit Study(int64_t seed = 0, Sampler sampler = Sampler::TPE) : rng_(seed), sampler_(sampler) {}
    void Optimize(const std::function<double(Trial &)> &objective, int nTrials, const volatile int32_t *cancel = nullptr) {
        for (int t = 0; t < nTrials; t++) {
            if (cancel && *cancel) throw std::runtime_error("context canceled");  // study.WithContext(ctx)
            double v;
            std::map<std::string, double> drew;
            if (sampler_ == Sampler::TPE) {
                TpeTrial trial(rng_, done_);
                v = objective(trial);
                drew = trial.Values;
            } else {
                RandomTrial trial(rng_);
                v = objective(trial);
                drew = trial.Values;
            }
            if (values_.empty() || v > best_) best_ = v;
            values_.push_back(v);
            done_.push_back(tpe::Finished{v, drew});
        }
    }
    double GetBestValue() const {
        if (values_.empty()) throw std::runtime_error("no trials");
        return best_;
    }
    const std::vector<double> &Values() const { return values_; }
    const std::vector<tpe::Finished> &Trials() const { return done_; }

   private:
    util::RandomGenerator rng_;
    Sampler sampler_;
    std::vector<double> values_;
    std::vector<tpe::Finished> done_;
    double best_ = 0;
};

void MarshalModel(std::ostream &w, const MatrixFactorization &m);               // model.go:320-328
std::unique_ptr<MatrixFactorization> UnmarshalModel(std::istream &r);            // model.go:330-350
}  // namespace cf

// ------------------------------------------------------------------------------------------
namespace ann {
using Result = std::vector<std::pair<int, float>>;  // []lo.Tuple2[int, float32]
// ann.Index (ann.go:21-25)
class Index {
   public:
    virtual ~Index() = default;
    virtual int Add(const std::vector<float> &v) = 0;
    virtual Result SearchIndex(int q, int k, bool prune0) = 0;  // throws std::out_of_range("index out of range")
    virtual Result SearchVector(const std::vector<float> &q, int k, bool prune0) = 0;
};
// ann.Bruteforce (bruteforce.go:24-83) with the scan on the GPU.  distance: GORSE_METRIC_*.
class Bruteforce : public Index {
   public:
    explicit Bruteforce(int metric, int device = 0) : metric_(metric), device_(device) {}
    ~Bruteforce() override {
        if (h_) gorse_topk_destroy(h_);
    }
    int Add(const std::vector<float> &v) override {  // returns len(vectors) like the reference (1-based)
        if (d_ == 0) d_ = (int)v.size();
        if ((int)v.size() != d_) throw std::invalid_argument("floats: slice lengths do not match");
        data_.insert(data_.end(), v.begin(), v.end());
        dirty_ = true;
        return (int)(data_.size() / (size_t)d_);
    }
    int Len() const { return d_ ? (int)(data_.size() / (size_t)d_) : 0; }
    Result SearchIndex(int q, int k, bool prune0) override {
        if (q < 0 || q >= Len()) throw std::out_of_range("index out of range: " + std::to_string(q));
        sync();
        int64_t qq = q;
        std::vector<int32_t> idx((size_t)k);
        std::vector<float> dist((size_t)k);
        int32_t cnt = 0;
        check(gorse_topk_search_index(h_, &qq, 1, k, prune0, idx.data(), dist.data(), &cnt));
        Result r;
        for (int t = 0; t < cnt; t++) r.emplace_back(idx[(size_t)t], dist[(size_t)t]);
        return r;
    }
    Result SearchVector(const std::vector<float> &q, int k, bool prune0) override {
        Result r;
        if (Len() == 0) return r;
        if ((int)q.size() != d_) throw std::invalid_argument("floats: slice lengths do not match");
        sync();
        std::vector<int32_t> idx((size_t)k);
        std::vector<float> dist((size_t)k);
        int32_t cnt = 0;
        check(gorse_topk_search_vector(h_, q.data(), 1, k, prune0, idx.data(), dist.data(), &cnt));
        for (int t = 0; t < cnt; t++) r.emplace_back(idx[(size_t)t], dist[(size_t)t]);
        return r;
    }
    // bulk form: SearchIndex for every stored vector (item-to-item build)
    void SearchAll(int k, std::vector<int32_t> &idx, std::vector<float> &dist) {
        sync();
        idx.resize((size_t)Len() * (size_t)k);
        dist.resize((size_t)Len() * (size_t)k);
        check(gorse_topk_all_pairs(h_, 0, Len(), k, idx.data(), dist.data()));
    }

   private:
    void sync() {
        if (!dirty_ && h_) return;
        if (h_) gorse_topk_destroy(h_);
        h_ = nullptr;
        check(gorse_topk_create(&h_, device_, Len(), d_, GORSE_DTYPE_F32, metric_, data_.data()));
        dirty_ = false;
    }
    int metric_, device_, d_ = 0;
    bool dirty_ = false;
    std::vector<float> data_;
    gorse_topk *h_ = nullptr;
};
}  // namespace ann

}  // namespace gorse
