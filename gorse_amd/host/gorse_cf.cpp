// gorse_cf.cpp -- implementation of the host mirror (see gorse_cf.hpp).  Numeric work goes through
// the C ABI only; this file is the C++ twin of what model/cf/bpr_hip.go / als_hip.go would contain.
#include <chrono>

#include "gorse_cf.hpp"

#include "gob.hpp"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <sstream>

namespace gorse {
namespace cf {

namespace {
void flatten(const std::vector<std::vector<int32_t>> &rows, size_t nrows, std::vector<int64_t> &ptr,
             std::vector<int32_t> &idx) {
    ptr.assign(nrows + 1, 0);
    for (size_t r = 0; r < nrows; r++) ptr[r + 1] = ptr[r] + (r < rows.size() ? (int64_t)rows[r].size() : 0);
    idx.resize((size_t)ptr[nrows]);
    for (size_t r = 0; r < nrows && r < rows.size(); r++) std::copy(rows[r].begin(), rows[r].end(), idx.begin() + ptr[r]);
}
std::string fmt(const char *f, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof(buf), f, ap);
    va_end(ap);
    return buf;
}
// protobuf varint / LatentFactor{1: string id, 2: packed float data}
void put_varint(std::ostream &w, uint64_t v) {
    while (v >= 0x80) {
        w.put((char)(v | 0x80));
        v >>= 7;
    }
    w.put((char)v);
}
uint64_t get_varint(std::istream &r) {
    uint64_t v = 0;
    int shift = 0;
    for (;;) {
        int c = r.get();
        if (c < 0) throw std::runtime_error("unexpected EOF");
        if (shift > 63) throw std::runtime_error("varint too long");
        v |= (uint64_t)(c & 0x7f) << shift;
        if (!(c & 0x80)) break;
        shift += 7;
    }
    return v;
}
// bytes left in a seekable stream (a length field of a corrupt file must not size an allocation); "unknown" = huge
uint64_t remaining(std::istream &r) {
    const auto at = r.tellg();
    if (at < 0) return UINT64_MAX;
    r.seekg(0, std::ios::end);
    const auto end = r.tellg();
    r.seekg(at);
    return end < at ? 0 : (uint64_t)(end - at);
}
size_t varint_len(uint64_t v) {
    size_t n = 1;
    while (v >= 0x80) {
        v >>= 7;
        n++;
    }
    return n;
}
void write_latent(std::ostream &w, const std::string &id, const float *data, int d) {
    size_t body = 0;
    if (!id.empty()) body += 1 + varint_len(id.size()) + id.size();
    if (d > 0) body += 1 + varint_len((uint64_t)d * 4) + (size_t)d * 4;
    put_varint(w, body);  // pbutil.WriteDelimited
    if (!id.empty()) {
        w.put(0x0A);
        put_varint(w, id.size());
        w.write(id.data(), (std::streamsize)id.size());
    }
    if (d > 0) {
        w.put(0x12);
        put_varint(w, (uint64_t)d * 4);
        w.write((const char *)data, (std::streamsize)d * 4);
    }
}
void read_latent(std::istream &r, std::string &id, std::vector<float> &data) {
    uint64_t body = get_varint(r);
    if (body > remaining(r)) throw std::runtime_error("unexpected EOF");
    std::string buf((size_t)body, '\0');
    r.read(&buf[0], (std::streamsize)body);
    if ((uint64_t)r.gcount() != body) throw std::runtime_error("unexpected EOF");
    std::istringstream s(buf);
    id.clear();
    data.clear();
    while (s.peek() != EOF) {
        uint64_t tag = get_varint(s);
        if (tag == 0x0A) {
            uint64_t n = get_varint(s);
            if (n > remaining(s)) throw std::runtime_error("unexpected EOF");
            id.resize((size_t)n);
            s.read(&id[0], (std::streamsize)n);
        } else if (tag == 0x12) {
            uint64_t n = get_varint(s);
            if (n > remaining(s) || n % 4) throw std::runtime_error("bad packed float field");
            data.resize((size_t)n / 4);
            s.read((char *)data.data(), (std::streamsize)n);
        } else if (tag == 0x15) {  // unpacked float
            float f;
            s.read((char *)&f, 4);
            if (s.gcount() != 4) throw std::runtime_error("unexpected EOF");
            data.push_back(f);
        } else {
            throw std::runtime_error("unknown LatentFactor field");
        }
    }
}
template <typename T>
void put_le(std::ostream &w, T v) {
    w.write((const char *)&v, sizeof(T));
}
template <typename T>
T get_le(std::istream &r) {
    T v;
    r.read((char *)&v, sizeof(T));
    if (r.gcount() != (std::streamsize)sizeof(T)) throw std::runtime_error("unexpected EOF");
    return v;
}
}  // namespace

void MatrixFactorization::Init(const dataset::Dataset &trainSet) {
    UserIndex = trainSet.GetUserDict();
    ItemIndex = trainSet.GetItemDict();
    const auto &uf = trainSet.GetUserFeedback();
    const auto &itf = trainSet.GetItemFeedback();
    UserPredictable.assign((size_t)trainSet.CountUsers(), false);
    for (size_t u = 0; u < UserPredictable.size(); u++) UserPredictable[u] = u < uf.size() && !uf[u].empty();
    ItemPredictable.assign((size_t)trainSet.CountItems(), false);
    for (size_t i = 0; i < ItemPredictable.size(); i++) ItemPredictable[i] = i < itf.size() && !itf[i].empty();
}

void MatrixFactorization::create_handle(const dataset::Dataset &trainSet, bool with_items, int device, ResidentDataset *res) {
    release();
    device_ = device;
    if (res) {  // borrow the resident copy when it is this very training set at this nFactors
        const bool same = res->h && res->trainSet == (const void *)&trainSet && res->U == trainSet.CountUsers() &&
                          res->I == trainSet.CountItems() && res->N == trainSet.CountFeedback() && res->nFactors == nFactors_ &&
                          res->device == device && (res->with_items || !with_items);
        if (!same) {
            if (res->h) gorse_mf_destroy(res->h);
            res->h = nullptr;
            std::vector<int64_t> uptr, iptr;
            std::vector<int32_t> uidx, iidx;
            flatten(trainSet.GetUserFeedback(), (size_t)trainSet.CountUsers(), uptr, uidx);
            flatten(trainSet.GetItemFeedback(), (size_t)trainSet.CountItems(), iptr, iidx);  // always: ALS may come next
            check(gorse_mf_create(&res->h, device, trainSet.CountUsers(), trainSet.CountItems(), nFactors_, uptr.data(),
                                  uidx.data(), iptr.data(), iidx.data()));
            res->trainSet = (const void *)&trainSet;
            res->U = trainSet.CountUsers(), res->I = trainSet.CountItems(), res->N = trainSet.CountFeedback();
            res->nFactors = nFactors_, res->device = device, res->with_items = true;
            res->uploads++;
        } else {
            res->reuses++;
        }
        h_ = res->h;
        borrowed_ = true;
        handle_train_ = (const void *)&trainSet;
        resident_eval_ = nullptr;  // a lent handle: whatever lists it holds are not this Fit's
        resident_gen_ = 0;
        check(gorse_mf_set_factors(h_, UserFactor.data(), ItemFactor.data()));
        return;
    }
    std::vector<int64_t> uptr, iptr;
    std::vector<int32_t> uidx, iidx;
    flatten(trainSet.GetUserFeedback(), (size_t)trainSet.CountUsers(), uptr, uidx);
    if (with_items) flatten(trainSet.GetItemFeedback(), (size_t)trainSet.CountItems(), iptr, iidx);
    check(gorse_mf_create(&h_, device, trainSet.CountUsers(), trainSet.CountItems(), nFactors_, uptr.data(), uidx.data(),
                          with_items ? iptr.data() : nullptr, with_items ? iidx.data() : nullptr));
    handle_train_ = (const void *)&trainSet;
    check(gorse_mf_set_factors(h_, UserFactor.data(), ItemFactor.data()));
}

void MatrixFactorization::ensure_resident() {
    if (h_) return;
    if (Invalid()) throw std::runtime_error("model is not fitted");
    // a model restored by Unmarshal has factors but no dataset: an empty feedback structure is enough for scoring
    const int64_t U = (int64_t)(UserFactor.size() / (size_t)nFactors_), I = (int64_t)(ItemFactor.size() / (size_t)nFactors_);
    std::vector<int64_t> uptr((size_t)U + 1, 0);
    int32_t dummy = 0;
    check(gorse_mf_create(&h_, device_, U, I, nFactors_, uptr.data(), &dummy, nullptr, nullptr));
    check(gorse_mf_set_factors(h_, UserFactor.data(), ItemFactor.data()));
}

std::vector<float> Evaluate(MatrixFactorization &estimator, dataset::Dataset &testSet, dataset::Dataset &trainSet, int topK,
                            int numCandidates, int nJobs, const std::vector<Metric> &scorers) {
    (void)nJobs;  // the device ranks every user at once; sums are taken in user order (== nJobs 1)
    // a production split has no preloaded negatives: sample them on the device when the handle of this Fit holds trainSet
    // (the host loop of Dataset::SampleUserNegatives draws the same lists, only user after user)
    if (!testSet.HasNegatives()) estimator.SampleNegativesOnDevice(testSet, trainSet, numCandidates);
    if (estimator.HasResidentCandidates(testSet, numCandidates)) {  // every later Evaluate of the Fit: no upload, rank lists only
        const auto &tf = testSet.GetUserFeedback();
        std::vector<int32_t> users;
        auto ranks = estimator.RankResident(users, topK);
        std::vector<float> sum(scorers.size(), 0.0f);
        float count = 0;
        for (size_t t = 0; t < users.size(); t++) {
            const auto &row = tf[(size_t)users[t]];
            TargetSet target(row.begin(), row.end());
            count++;
            for (size_t m = 0; m < scorers.size(); m++) sum[m] += scorers[m](target, ranks[t]);
        }
        const float inv = 1 / count;
        for (auto &x : sum) x *= inv;
        return sum;
    }
    const auto &negatives = testSet.SampleUserNegatives(trainSet, numCandidates);
    const auto &tf = testSet.GetUserFeedback();
    std::vector<int32_t> users;
    std::vector<std::vector<int32_t>> cands;
    std::vector<TargetSet> targets;
    for (int u = 0; u < testSet.CountUsers(); u++) {
        if ((size_t)u >= tf.size()) break;
        TargetSet t(tf[(size_t)u].begin(), tf[(size_t)u].end());
        if (t.empty()) continue;
        std::vector<int32_t> c(tf[(size_t)u]);
        if ((size_t)u < negatives.size()) c.insert(c.end(), negatives[(size_t)u].begin(), negatives[(size_t)u].end());
        users.push_back(u);
        cands.push_back(std::move(c));
        targets.push_back(std::move(t));
    }
    std::vector<float> sum(scorers.size(), 0.0f);
    float count = 0;
    if (!users.empty()) {
        auto ranks = estimator.RankMany(users, cands, topK);
        for (size_t t = 0; t < users.size(); t++) {
            count++;
            for (size_t m = 0; m < scorers.size(); m++) sum[m] += scorers[m](targets[t], ranks[t]);
        }
    }
    const float inv = 1 / count;
    for (auto &s : sum) s *= inv;  // floats.MulConst(sum, 1/count)
    return sum;
}

Score MatrixFactorization::fit_loop(const char *tag, int nEpochs, dataset::Dataset &trainSet, dataset::Dataset &valSet,
                                    const FitConfig &config, const std::function<int32_t(int)> &run_epoch) {
    auto log = [&](const std::string &s) {
        if (config.Log) config.Log(s);
    };
    const std::vector<Metric> metrics{NDCG, Precision, Recall};
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    // eval_time / fit_time as in the reference's log lines (model.go:432-440, 496-503).  Epochs between two evaluations are
    // only enqueued (see BPR::Fit), so fit_time is the mean over the epochs since the last evaluation, not the last epoch's --
    // and, where the library timed the epochs on the device (gorse_mf_epoch_times: hipEvents on the update stream, BPR), the mean of
    // THEIR device times: what an enqueueing host measures with its own clock is when it issued an epoch, not how long it ran.
    auto t_eval = clk::now();
    auto score = Evaluate(*this, valSet, trainSet, config.TopK, config.Candidates, config.Jobs, metrics);
    std::vector<std::pair<int, float>> scores{{0, score[0]}};
    log(fmt("fit %s 0/%d eval_time=%.3fms NDCG@%d=%g Precision@%d=%g Recall@%d=%g", tag, nEpochs, ms_since(t_eval), config.TopK,
            score[0], config.TopK, score[1], config.TopK, score[2]));
    auto t_fit = clk::now();
    int fit_epochs = 0;
    epochs_done_ = 0;
    if (h_) (void)gorse_mf_epoch_times(h_, nullptr, nullptr, nullptr, 1);  // epochs of an earlier Fit on a lent handle do not count
    for (int epoch = 1; epoch <= nEpochs; epoch++) {
        int32_t rc = run_epoch(epoch);
        fit_epochs++;
        if (rc == GORSE_ERR_CANCELLED) {  // "fit bpr canceled" -> Score{} (model.go:490-493)
            log(fmt("fit %s canceled epoch=%d", tag, epoch));
            pull_factors();
            if (borrowed_) release();
            return Score{};
        }
        check(rc);
        if (epoch % config.Verbose == 0 || epoch == nEpochs) {
            double fit_ms = ms_since(t_fit) / fit_epochs;
            {
                int64_t timed = 0;
                double dev_ms = 0;
                if (h_ && gorse_mf_epoch_times(h_, &timed, &dev_ms, nullptr, 1) == GORSE_OK && timed == fit_epochs) fit_ms = dev_ms / timed;
            }
            t_eval = clk::now();
            score = Evaluate(*this, valSet, trainSet, config.TopK, config.Candidates, config.Jobs, metrics);
            scores.emplace_back(epoch, score[0]);
            log(fmt("fit %s %d/%d fit_time=%.3fms eval_time=%.3fms NDCG@%d=%g Precision@%d=%g Recall@%d=%g", tag, epoch, nEpochs,
                    fit_ms, ms_since(t_eval), config.TopK, score[0], config.TopK, score[1], config.TopK, score[2]));
            t_fit = clk::now();
            fit_epochs = 0;
            if (config.Patience > 0 && epoch > config.Patience) {
                // lo.MaxBy with strict > : the FIRST maximum
                auto best = scores[0];
                for (auto &s : scores)
                    if (s.second > best.second) best = s;
                if (best.first <= epoch - config.Patience) {
                    log(fmt("early stopping best_epoch=%d best_NDCG=%g patience=%d", best.first, best.second, config.Patience));
                    break;
                }
            }
        }
        epochs_done_ = epoch;  // (where the progress hook fires: a cancelled epoch and the epoch an early stop breaks at do not count)
        if (config.OnEpoch) config.OnEpoch(epoch);
    }
    const auto t_pull = clk::now();
    pull_factors();  // the reference's [][]float32 rows, before Marshal / GetUserFactor are used
    if (borrowed_) release();  // a lent handle goes back: the next Fit overwrites its factors (Predict re-uploads ours)
    log(fmt("fit %s teardown pull=%.3fms", tag, ms_since(t_pull)));
    log(fmt("fit %s complete NDCG@%d=%g", tag, config.TopK, score[0]));
    return Score{score[0], score[1], score[2]};
}

Score BPR::Fit(dataset::Dataset &trainSet, dataset::Dataset &valSet, const FitConfig &config) {
    // Init (model.go:532-540): users first, then items
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = clk::now();
    GetRandomGenerator().NormalMatrix(trainSet.CountUsers(), nFactors_, initMean, initStdDev, UserFactor);
    GetRandomGenerator().NormalMatrix(trainSet.CountItems(), nFactors_, initMean, initStdDev, ItemFactor);
    const auto t1 = clk::now();
    Init(trainSet);
    const auto t2 = clk::now();
    create_handle(trainSet, false, config.Device, config.Resident);
    // (not a line of the reference's log: where a Fit's time goes outside its epochs and evaluations)
    if (config.Log) config.Log(fmt("fit bpr setup draws=%.3fms init=%.3fms handle=%.3fms", ms(t0, t1), ms(t1, t2), ms(t2, clk::now())));
    // per-Fit sampler seed, as rng[i] = NewRandomGenerator(bpr.GetRandomGenerator().Int63()) (model.go:420-423)
    const uint64_t seed = (uint64_t)GetRandomGenerator().Int63();
    // Jobs <= 1: parallel.Parallel runs the samples strictly in order (parallel.go:34-43) -> sequential
    // schedule; Jobs > 1: Hogwild workers -> the atomic Hogwild schedule.
    // Jobs > 1 is the reference's Hogwild: workers write item rows without a lock (model.go:478-488) -- GORSE_BPR_HOGWILD_STORES keeps
    // that semantics for the cold negatives and atomics everywhere else; GORSE_BPR_HOGWILD_ATOMIC (no lost item update) is the
    // caller's choice through the C ABI, not Fit's
    const int mode = config.Jobs <= 1 ? GORSE_BPR_SEQUENTIAL : GORSE_BPR_HOGWILD_STORES;
    const int64_t n = trainSet.CountFeedback();
    // Between two evaluations the epochs are only enqueued: the sampler and the counting sort of epoch e + 1 run under the
    // update kernel of epoch e.  The epoch in front of an evaluation (and every sequential epoch) is the synchronous call,
    // which also polls the cancel flag.
    return fit_loop("bpr", nEpochs, trainSet, valSet, config, [&](int epoch) {
        const bool eval_next = epoch % config.Verbose == 0 || epoch == nEpochs;
        const bool cancelled = config.Cancel && *config.Cancel;
        // (OnEpoch = the reference's span.Add(1), a progress count: it fires when the epoch has been ISSUED, as in the Go twin,
        // integration/go/model/cf/bpr_hip.go:93 -- until round 5 a hook forced the synchronous entry point for every epoch here)
        if (!eval_next && mode != GORSE_BPR_SEQUENTIAL && !cancelled) {
            // at most two epochs in flight: the preparation of epoch e + 1 still runs under the update of epoch e, and a cancelled
            // context is seen within two epochs (the wait looks at the flag every 20 us) instead of at the next evaluation
            const int32_t rc = gorse_mf_epoch_throttle(h_, kEnqueueDepth, config.Cancel);
            if (rc != GORSE_OK) {
                if (rc == GORSE_ERR_CANCELLED) (void)gorse_mf_synchronize(h_);  // drain what is in flight: the factors are pulled next
                return rc;
            }
            return gorse_bpr_epoch_enqueue(h_, n, lr, reg, seed, (uint64_t)epoch, 0, mode);
        }
        return gorse_bpr_epoch(h_, n, lr, reg, seed, (uint64_t)epoch, 0, mode, config.Cancel, nullptr);
    });
}

Score ALS::Fit(dataset::Dataset &trainSet, dataset::Dataset &valSet, const FitConfig &config) {
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = clk::now();
    GetRandomGenerator().NormalMatrix(trainSet.CountUsers(), nFactors_, initMean, initStdDev, UserFactor);
    GetRandomGenerator().NormalMatrix(trainSet.CountItems(), nFactors_, initMean, initStdDev, ItemFactor);
    const auto t1 = clk::now();
    Init(trainSet);
    const auto t2 = clk::now();
    create_handle(trainSet, true, config.Device, config.Resident);
    if (config.Log) config.Log(fmt("fit als setup draws=%.3fms init=%.3fms handle=%.3fms", ms(t0, t1), ms(t1, t2), ms(t2, clk::now())));
    return fit_loop("als", nEpochs, trainSet, valSet, config,
                    [&](int) { return gorse_als_epoch(h_, weight, reg, config.Cancel); });
}

void MatrixFactorization::Marshal(std::ostream &w) const {
    // encoding.WriteGob(w, baseModel.Params) (model.go:208, encoding.go:98-107): int32 byte count + the gob stream of
    // map[ParamName]any.  The reference reads NFactors / NEpochs with GetInt (only `int` matches, params.go:86-97) and
    // RandomState with GetInt64 (int64 or int), everything else through GetFloat32 (float32 / float64 / int).
    gob::Entries entries;
    for (auto &kv : Params) {
        const bool integral = kv.first == model::NFactors || kv.first == model::NEpochs || kv.first == model::RandomState;
        entries.emplace_back(kv.first, integral && kv.second == (double)(int64_t)kv.second ? gob::Value::of_int((int64_t)kv.second)
                                                                                            : gob::Value::of_float(kv.second));
    }
    const std::string params = gob::encode_map("Params", entries);
    put_le<int32_t>(w, (int32_t)params.size());
    w.write(params.data(), (std::streamsize)params.size());
    int64_t cnt = 0;
    for (int32_t u = 0; u < UserIndex->Count(); u++) cnt += IsUserPredictable(u);
    put_le<int64_t>(w, cnt);
    std::string id;
    for (int32_t u = 0; u < UserIndex->Count(); u++)
        if (IsUserPredictable(u)) {
            UserIndex->String(u, id);
            write_latent(w, id, GetUserFactor(u), nFactors_);
        }
    cnt = 0;
    for (int32_t i = 0; i < ItemIndex->Count(); i++) cnt += IsItemPredictable(i);
    put_le<int64_t>(w, cnt);
    for (int32_t i = 0; i < ItemIndex->Count(); i++)
        if (IsItemPredictable(i)) {
            ItemIndex->String(i, id);
            write_latent(w, id, GetItemFactor(i), nFactors_);
        }
}

void MatrixFactorization::Unmarshal(std::istream &r) {
    release();
    model::Params p;  // encoding.ReadGob(r, &baseModel.Params) (model.go:251)
    const int32_t nbytes = get_le<int32_t>(r);
    if (nbytes < 0 || (uint64_t)nbytes > remaining(r)) throw std::runtime_error("bad gob length");
    std::string params((size_t)nbytes, '\0');
    r.read(&params[0], nbytes);
    if (r.gcount() != (std::streamsize)nbytes) throw std::runtime_error("unexpected EOF");
    for (auto &kv : gob::decode_map(params))
        if (kv.second.kind != gob::Value::String) p[kv.first] = kv.second.number();
    SetParams(p);
    auto read_side = [&](std::shared_ptr<dataset::FreqDict> &dict, std::vector<bool> &pred, std::vector<float> &fac) {
        int64_t cnt = get_le<int64_t>(r);
        // a record holds its factors (4 bytes each): a count the rest of the file cannot hold is corrupt
        if (cnt < 0 || (uint64_t)cnt > remaining(r) || nFactors_ <= 0 || (uint64_t)cnt * (uint64_t)nFactors_ > remaining(r))
            throw std::runtime_error("latent factor count does not fit the file");
        dict = std::make_shared<dataset::FreqDict>();
        pred.assign((size_t)cnt, false);
        fac.assign((size_t)cnt * (size_t)nFactors_, 0.0f);
        std::string id;
        std::vector<float> data;
        for (int64_t k = 0; k < cnt; k++) {
            read_latent(r, id, data);
            int32_t idx = dict->Add(id);
            pred[(size_t)idx] = true;
            if ((int)data.size() != nFactors_) throw std::runtime_error("latent factor length mismatch");
            std::copy(data.begin(), data.end(), fac.begin() + (ptrdiff_t)((size_t)idx * (size_t)nFactors_));
        }
    };
    read_side(UserIndex, UserPredictable, UserFactor);
    read_side(ItemIndex, ItemPredictable, ItemFactor);
}

void MarshalModel(std::ostream &w, const MatrixFactorization &m) {
    std::string name = m.Name();  // encoding.WriteString: LE int32 length + bytes
    put_le<int32_t>(w, (int32_t)name.size());
    w.write(name.data(), (std::streamsize)name.size());
    m.Marshal(w);
}

std::unique_ptr<MatrixFactorization> UnmarshalModel(std::istream &r) {
    int32_t len = get_le<int32_t>(r);
    if (len < 0 || (uint64_t)len > remaining(r)) throw std::runtime_error("bad model name length");
    std::string name((size_t)len, '\0');
    r.read(&name[0], len);
    std::unique_ptr<MatrixFactorization> m;
    if (name == "bpr")
        m.reset(new BPR());
    else if (name == "als")
        m.reset(new ALS());
    else
        throw std::runtime_error("unknown model " + name);
    m->Unmarshal(r);
    return m;
}

}  // namespace cf
}  // namespace gorse
