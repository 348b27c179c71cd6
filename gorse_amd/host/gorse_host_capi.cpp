// gorse_host_capi.cpp -- flat C entry points over the C++ host mirror, so that the parity tests
// (pytest + ctypes) can drive gorse::cf::BPR / ALS / Evaluate / ann::Bruteforce / heap::* exactly as
// the reference's Go tests drive their Go twins.  Test/binding plumbing only.
#include <cstring>
#include <sstream>

#include "gorse_cf.hpp"
#include "gorse_vectors.hpp"
#include "../csrc/rank_keys.hpp"
#include "../csrc/sparse_host.hpp"
#include "../csrc/als_plan.hpp"
#include "../csrc/bpr_bins.hpp"
#include "../csrc/topk_sym.hpp"

using namespace gorse;

namespace {
thread_local std::string g_err;
template <typename F>
int32_t guard(F &&f) {
    try {
        f();
        return 0;
    } catch (const HipError &e) {
        g_err = e.what();
        return e.code;
    } catch (const storage::ErrNotFound &e) {
        g_err = e.what();
        return -201;
    } catch (const storage::ErrAlreadyExists &e) {
        g_err = e.what();
        return -202;
    } catch (const storage::ErrNotSupported &e) {
        g_err = e.what();
        return -203;
    } catch (const std::out_of_range &e) {
        g_err = e.what();
        return GORSE_ERR_RANGE;
    } catch (const std::invalid_argument &e) {
        g_err = e.what();
        return GORSE_ERR_INVALID;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -100;
    }
}
struct FitLog {
    std::string text;
};
}  // namespace

extern "C" {
const char *gh_last_error() { return g_err.c_str(); }

// ---- dataset ---------------------------------------------------------------------------------
void *gh_dataset_new() { return new dataset::Dataset(); }
void *gh_dataset_new_shared(void *other) { return new dataset::Dataset(*(dataset::Dataset *)other, true); }
void gh_dataset_free(void *d) { delete (dataset::Dataset *)d; }
void gh_dataset_add_feedback(void *d, const int32_t *u, const int32_t *i, int64_t n) {
    auto *ds = (dataset::Dataset *)d;
    for (int64_t t = 0; t < n; t++) ds->AddFeedbackIndexed(u[t], i[t]);
}
void gh_dataset_add_feedback_str(void *d, const char *user, const char *item) {
    ((dataset::Dataset *)d)->AddFeedback(user, item);
}
void gh_dataset_add_user(void *d, const char *user) { ((dataset::Dataset *)d)->AddUser(user); }
void gh_dataset_add_item(void *d, const char *item) { ((dataset::Dataset *)d)->AddItem(item); }
void gh_dataset_set_negatives(void *d, int32_t user, const int32_t *neg, int32_t n) {
    ((dataset::Dataset *)d)->SetNegatives(user, std::vector<int32_t>(neg, neg + n));
}
int32_t gh_dataset_count_users(void *d) { return ((dataset::Dataset *)d)->CountUsers(); }
int32_t gh_dataset_count_items(void *d) { return ((dataset::Dataset *)d)->CountItems(); }
int32_t gh_dataset_count_feedback(void *d) { return ((dataset::Dataset *)d)->CountFeedback(); }
// SplitCF(numTestUsers, seed): two new datasets (free both)
int32_t gh_dataset_split_cf(void *d, int32_t num_test_users, int64_t seed, void **train, void **test) {
    return guard([&] {
        auto pr = ((dataset::Dataset *)d)->SplitCF(num_test_users, seed);
        *train = new dataset::Dataset(std::move(pr.first));
        *test = new dataset::Dataset(std::move(pr.second));
    });
}
// LoadDataFromBuiltIn's two files, given as text
int32_t gh_dataset_load_ncf(const char *train_txt, const char *test_txt, void **train, void **test) {
    return guard([&] {
        std::istringstream a(train_txt), b(test_txt);
        auto pr = dataset::Dataset::LoadNCF(a, b);
        *train = new dataset::Dataset(std::move(pr.first));
        *test = new dataset::Dataset(std::move(pr.second));
    });
}
// row `row` of GetUserFeedback (side 0) / GetItemFeedback (side 1) / the stored negatives (side 2); returns its length
int32_t gh_dataset_row(void *d, int32_t side, int32_t row, int32_t *out, int32_t cap) {
    auto *ds = (dataset::Dataset *)d;
    const auto &m = side == 0 ? ds->GetUserFeedback() : (side == 1 ? ds->GetItemFeedback() : ds->Negatives());
    if (row < 0 || (size_t)row >= m.size()) return 0;
    const auto &r = m[(size_t)row];
    for (size_t t = 0; t < r.size() && (int32_t)t < cap; t++) out[t] = r[t];
    return (int32_t)r.size();
}
// GetUserIDF (side 0) / GetItemIDF (side 1) into out[CountUsers() / CountItems() of the dictionary]
int32_t gh_dataset_idf(void *d, int32_t side, float *out, int32_t cap) {
    auto v = side == 0 ? ((dataset::Dataset *)d)->GetUserIDF() : ((dataset::Dataset *)d)->GetItemIDF();
    for (size_t t = 0; t < v.size() && (int32_t)t < cap; t++) out[t] = v[t];
    return (int32_t)v.size();
}

// ---- FreqDict / RandomGenerator (test hooks over dataset/dict.go, common/util/random.go) --------------------------------------
void *gh_freqdict_new() { return new dataset::FreqDict(); }
void gh_freqdict_free(void *d) { delete (dataset::FreqDict *)d; }
int32_t gh_freqdict_add(void *d, const char *s) { return ((dataset::FreqDict *)d)->Add(s); }
int32_t gh_freqdict_add_no_count(void *d, const char *s) { return ((dataset::FreqDict *)d)->AddNoCount(s); }
int32_t gh_freqdict_count(void *d) { return ((dataset::FreqDict *)d)->Count(); }
int32_t gh_freqdict_freq(void *d, int32_t id) { return ((dataset::FreqDict *)d)->Freq(id); }
int32_t gh_freqdict_id(void *d, const char *s) { return ((dataset::FreqDict *)d)->Id(s); }
void gh_rng_normal_matrix(int64_t seed, int64_t rows, int64_t cols, float mean, float stddev, float *out) {
    util::RandomGenerator rng(seed);
    std::vector<float> m;
    rng.NormalMatrix(rows, cols, mean, stddev, m);
    std::copy(m.begin(), m.end(), out);
}
// successive SampleInt32(low, high, n[i], exclude) calls on ONE generator, like random_test.go:51-59; out = concatenation
int32_t gh_rng_sample_int32(int64_t seed, int32_t low, int32_t high, const int32_t *ns, int32_t n_calls, const int32_t *exclude,
                            int32_t n_exclude, int32_t *out, int32_t *out_lens) {
    util::RandomGenerator rng(seed);
    const std::set<int32_t> ex(exclude, exclude + n_exclude);
    int32_t at = 0;
    for (int32_t c = 0; c < n_calls; c++) {
        auto v = rng.SampleInt32(low, high, ns[c], ex);
        out_lens[c] = (int32_t)v.size();
        for (int32_t x : v) out[at++] = x;
    }
    return at;
}

// ---- models ------------------------------------------------------------------------------------
static model::Params make_params(const char **names, const double *vals, int32_t n) {
    model::Params p;
    for (int32_t k = 0; k < n; k++) p[names[k]] = vals[k];
    return p;
}
void *gh_bpr_new(const char **names, const double *vals, int32_t n) { return new cf::BPR(make_params(names, vals, n)); }
void *gh_als_new(const char **names, const double *vals, int32_t n) { return new cf::ALS(make_params(names, vals, n)); }
void gh_model_free(void *m) { delete (cf::MatrixFactorization *)m; }
const char *gh_model_name(void *m) { return ((cf::MatrixFactorization *)m)->Name(); }

// Fit(ctx, trainSet, valSet, config) -> Score.  log_out receives the zap-style log lines.
int32_t gh_model_fit(void *m, void *train, void *val, int32_t jobs, int32_t verbose, int32_t candidates, int32_t topk,
                     int32_t patience, const volatile int32_t *cancel, float *score3, int32_t *epochs_done, char *log_out,
                     int64_t log_cap) {
    return guard([&] {
        cf::FitConfig c;
        c.Jobs = jobs, c.Verbose = verbose, c.Candidates = candidates, c.TopK = topk, c.Patience = patience, c.Cancel = cancel;
        std::string logs;
        // (the count of epochs comes from the model: EpochsDone)
        c.Log = [&](const std::string &s) { logs += s + "\n"; };
        cf::Score s = ((cf::MatrixFactorization *)m)->Fit(*(dataset::Dataset *)train, *(dataset::Dataset *)val, c);
        score3[0] = s.NDCG, score3[1] = s.Precision, score3[2] = s.Recall;
        if (epochs_done) *epochs_done = ((cf::MatrixFactorization *)m)->EpochsDone();
        if (log_out && log_cap > 0) {
            size_t n = std::min<size_t>((size_t)log_cap - 1, logs.size());
            memcpy(log_out, logs.data(), n);
            log_out[n] = 0;
        }
    });
}
int32_t gh_model_predict(void *m, const char *user, const char *item, float *out) {
    return guard([&] { *out = ((cf::MatrixFactorization *)m)->Predict(user, item); });
}
int32_t gh_model_internal_predict(void *m, int32_t u, int32_t i, float *out) {
    return guard([&] { *out = ((cf::MatrixFactorization *)m)->internalPredict(u, i); });
}
int32_t gh_model_n_factors(void *m) { return ((cf::MatrixFactorization *)m)->NFactors(); }
int32_t gh_model_count_users(void *m) {
    auto *mf = (cf::MatrixFactorization *)m;
    return mf->UserIndex ? mf->UserIndex->Count() : -1;
}
int32_t gh_model_count_items(void *m) {
    auto *mf = (cf::MatrixFactorization *)m;
    return mf->ItemIndex ? mf->ItemIndex->Count() : -1;
}
int32_t gh_model_user_index(void *m, const char *id) {
    auto *mf = (cf::MatrixFactorization *)m;
    return mf->UserIndex ? mf->UserIndex->Id(id) : -1;
}
int32_t gh_model_item_index(void *m, const char *id) {
    auto *mf = (cf::MatrixFactorization *)m;
    return mf->ItemIndex ? mf->ItemIndex->Id(id) : -1;
}
void gh_model_get_user_factor(void *m, int32_t u, float *out) {
    auto *mf = (cf::MatrixFactorization *)m;
    memcpy(out, mf->GetUserFactor(u), (size_t)mf->NFactors() * 4);
}
void gh_model_get_item_factor(void *m, int32_t i, float *out) {
    auto *mf = (cf::MatrixFactorization *)m;
    memcpy(out, mf->GetItemFactor(i), (size_t)mf->NFactors() * 4);
}
int32_t gh_model_is_user_predictable(void *m, int32_t u) { return ((cf::MatrixFactorization *)m)->IsUserPredictable(u); }
int32_t gh_model_is_item_predictable(void *m, int32_t i) { return ((cf::MatrixFactorization *)m)->IsItemPredictable(i); }
void gh_model_clear(void *m) { ((cf::MatrixFactorization *)m)->Clear(); }
int32_t gh_model_invalid(void *m) { return ((cf::MatrixFactorization *)m)->Invalid(); }
// MarshalModel into a caller buffer; returns bytes needed
int64_t gh_model_marshal(void *m, char *buf, int64_t cap) {
    std::ostringstream o;
    cf::MarshalModel(o, *(cf::MatrixFactorization *)m);
    const std::string &s = o.str();
    if (buf && cap >= (int64_t)s.size()) memcpy(buf, s.data(), s.size());
    return (int64_t)s.size();
}
void *gh_model_unmarshal(const char *buf, int64_t n) {
    void *r = nullptr;
    guard([&] {
        std::istringstream i(std::string(buf, (size_t)n));
        r = cf::UnmarshalModel(i).release();
    });
    return r;
}
// test helper: a model whose factors are given (ids "0".."n-1"), the equivalent of Unmarshal
int32_t gh_model_load_factors(void *m, int32_t U, int32_t I, const float *P, const float *Q) {
    return guard([&] {
        auto *mf = (cf::MatrixFactorization *)m;
        mf->Clear();
        const int d = mf->NFactors();
        mf->UserIndex = std::make_shared<dataset::FreqDict>();
        mf->ItemIndex = std::make_shared<dataset::FreqDict>();
        for (int32_t u = 0; u < U; u++) mf->UserIndex->Add(std::to_string(u));
        for (int32_t i = 0; i < I; i++) mf->ItemIndex->Add(std::to_string(i));
        mf->UserPredictable.assign((size_t)U, true);
        mf->ItemPredictable.assign((size_t)I, true);
        mf->UserFactor.assign(P, P + (size_t)U * (size_t)d);
        mf->ItemFactor.assign(Q, Q + (size_t)I * (size_t)d);
    });
}
// Evaluate(estimator, testSet, trainSet, topK, numCandidates, nJobs, metrics...); metric ids 0..5
int32_t gh_evaluate(void *m, void *test, void *train, int32_t topk, int32_t candidates, int32_t jobs, const int32_t *metric_ids,
                    int32_t n_metrics, float *out) {
    return guard([&] {
        static const cf::Metric table[] = {cf::NDCG, cf::Precision, cf::Recall, cf::HR, cf::MAP, cf::MRR};
        std::vector<cf::Metric> ms;
        for (int32_t k = 0; k < n_metrics; k++) ms.push_back(table[metric_ids[k]]);
        auto r = cf::Evaluate(*(cf::MatrixFactorization *)m, *(dataset::Dataset *)test, *(dataset::Dataset *)train, topk,
                              candidates, jobs, ms);
        for (int32_t k = 0; k < n_metrics; k++) out[k] = r[(size_t)k];
    });
}
float gh_metric(int32_t id, const int32_t *target, int32_t nt, const int32_t *rank, int32_t nr) {
    static const cf::Metric table[] = {cf::NDCG, cf::Precision, cf::Recall, cf::HR, cf::MAP, cf::MRR};
    return table[id](cf::TargetSet(target, target + nt), std::vector<int32_t>(rank, rank + nr));
}

// ---- ModelSearch (optimize.go) ---------------------------------------------------------------------------------------
namespace {
// optimize_test.go:29-99: a model whose Fit returns NDCG = NFactors + InitMean + InitStdDev over a 4 x 4 x 1 grid
struct MockForSearch : cf::MatrixFactorization {
    const char *Name() const override { return "mock"; }
    cf::Score Fit(dataset::Dataset &, dataset::Dataset &, const cf::FitConfig &) override {
        cf::Score s;
        s.NDCG = Params.GetFloat32(model::NFactors, 0) + Params.GetFloat32(model::InitMean, 0) + Params.GetFloat32(model::InitStdDev, 0);
        return s;
    }
    model::Params SuggestParams(cf::Trial &trial) override {
        model::Params p;
        p[model::NFactors] = trial.SuggestDiscreteFloat(model::NFactors, 1, 4, 1);
        p[model::InitMean] = trial.SuggestDiscreteFloat(model::InitMean, 1, 4, 1);
        p[model::InitStdDev] = trial.SuggestDiscreteFloat(model::InitStdDev, 4, 4, 1);
        return p;
    }
};
void report(const cf::SearchResult &r, char *type_out, int64_t type_cap, char *params_out, int64_t params_cap, float *score3) {
    if (type_out && type_cap > (int64_t)r.Type.size()) std::memcpy(type_out, r.Type.c_str(), r.Type.size() + 1);
    std::string text;
    for (auto &kv : r.Params) {
        char num[64];
        snprintf(num, sizeof(num), "%.17g", kv.second);
        text += kv.first + "=" + num + "\n";
    }
    if (params_out && params_cap > (int64_t)text.size()) std::memcpy(params_out, text.c_str(), text.size() + 1);
    score3[0] = r.Score_.NDCG, score3[1] = r.Score_.Precision, score3[2] = r.Score_.Recall;
}
}  // namespace
// TestTPE's search (optimize_test.go:101-126); sampler: 0 = TPE (what the reference's test and master use), 1 = random;
// best_value = study.GetBestValue()
int32_t gh_search_mock(int32_t n_trials, int64_t seed, int32_t sampler, double *best_value, char *type_out, int64_t type_cap,
                       char *params_out, int64_t params_cap, float *score3) {
    return guard([&] {
        cf::ModelSearch search({{"mock", [] { return std::unique_ptr<cf::MatrixFactorization>(new MockForSearch()); }}}, nullptr,
                               nullptr, cf::FitConfig(), false);
        cf::Study study(seed, sampler ? cf::Study::Sampler::Random : cf::Study::Sampler::TPE);
        study.Optimize([&](cf::Trial &t) { return search.Objective(t); }, n_trials);
        *best_value = study.GetBestValue();
        report(search.Result(), type_out, type_cap, params_out, params_cap, score3);
    });
}
// optimizeCollaborativeFiltering (master/tasks.go:1268-1316): BPR and ALS with their default parameters (+ the overrides
// given, e.g. NEpochs), FitConfig{Jobs, Patience}, n_trials trials; counters = {dataset uploads, resident reuses, trials run}
int32_t gh_model_search(void *train, void *val, int32_t n_trials, int64_t seed, int32_t jobs, int32_t patience,
                        const char **names, const double *vals, int32_t n_over, int32_t keep_resident, const volatile int32_t *cancel,
                        char *type_out, int64_t type_cap, char *params_out, int64_t params_cap, float *score3, int32_t *counters) {
    return guard([&] {
        // the overrides are re-applied after SuggestParams (which replaces the whole parameter set, optimize.go:69)
        const model::Params over = make_params(names, vals, n_over);
        struct BPRo : cf::BPR {
            model::Params over;
            model::Params SuggestParams(cf::Trial &t) override {
                model::Params p = cf::BPR::SuggestParams(t);
                for (auto &kv : over) p[kv.first] = kv.second;
                return p;
            }
        };
        struct ALSo : cf::ALS {
            model::Params over;
            model::Params SuggestParams(cf::Trial &t) override {
                model::Params p = cf::ALS::SuggestParams(t);
                for (auto &kv : over) p[kv.first] = kv.second;
                return p;
            }
        };
        cf::FitConfig c;
        c.SetJobs(jobs).SetPatience(patience);
        c.Cancel = cancel;
        cf::ModelSearch search({{"BPR", [&] { auto m = new BPRo(); m->over = over; return std::unique_ptr<cf::MatrixFactorization>(m); }},
                                {"ALS", [&] { auto m = new ALSo(); m->over = over; return std::unique_ptr<cf::MatrixFactorization>(m); }}},
                               (dataset::Dataset *)train, (dataset::Dataset *)val, c, keep_resident != 0);
        int trials = 0;
        search.OnTrial = [&] { trials++; };
        cf::Study study(seed);
        study.Optimize([&](cf::Trial &t) { return search.Objective(t); }, n_trials, cancel);
        report(search.Result(), type_out, type_cap, params_out, params_cap, score3);
        if (counters) counters[0] = search.Resident().uploads, counters[1] = search.Resident().reuses, counters[2] = trials;
    });
}

// ---- the TPE sampler's pieces (tpe.hpp), for the CPU tests ---------------------------------------------------------------
// the Parzen estimator of `n` observations on [low, high]: returns the number of components (n + 1), fills weights / mus / sigmas
int32_t gh_tpe_parzen(const double *obs, int32_t n, double low, double high, double *weights, double *mus, double *sigmas, int32_t cap) {
    const cf::tpe::ParzenEstimator pe(std::vector<double>(obs, obs + n), low, high);
    const int32_t m = (int32_t)pe.mus.size();
    for (int32_t i = 0; i < m && i < cap; i++) weights[i] = pe.weights[(size_t)i], mus[i] = pe.mus[(size_t)i], sigmas[i] = pe.sigmas[(size_t)i];
    return m;
}
int32_t gh_tpe_gamma(int32_t n) { return cf::tpe::default_gamma(n); }
void gh_tpe_weights(int32_t n, double *out) {
    const std::vector<double> w = cf::tpe::default_weights(n);
    for (size_t i = 0; i < w.size(); i++) out[i] = w[i];
}
void gh_tpe_log_pdf(const double *samples, int32_t ns, const double *obs, int32_t n, double low, double high, double q, double *out) {
    const cf::tpe::ParzenEstimator pe(std::vector<double>(obs, obs + n), low, high);
    const std::vector<double> r = cf::tpe::gmm_log_pdf(std::vector<double>(samples, samples + ns), pe, low, high, q);
    for (size_t i = 0; i < r.size(); i++) out[i] = r[i];
}
// a study over one log-uniform parameter x in [low, high] with the objective -(log x - log target)^2: the values suggested
void gh_tpe_study_1d(int32_t n_trials, int64_t seed, int32_t sampler, double low, double high, double target, double *suggested) {
    cf::Study study(seed, sampler ? cf::Study::Sampler::Random : cf::Study::Sampler::TPE);
    int t = 0;
    study.Optimize([&](cf::Trial &trial) {
        const double x = trial.SuggestLogFloat("x", low, high);
        suggested[t++] = x;
        const double e = std::log(x) - std::log(target);
        return -e * e;
    }, n_trials);
}

// ---- gob (test hooks of gob.hpp) and the MatrixFactorizationUsers blob --------------------------------------------------
static int64_t copy_out(const std::string &s, char *buf, int64_t cap) {
    if (buf && cap >= (int64_t)s.size()) memcpy(buf, s.data(), s.size());
    return (int64_t)s.size();
}
int64_t gh_gob_encode_int(int64_t v, char *buf, int64_t cap) { return copy_out(gob::encode_int(v), buf, cap); }
int64_t gh_gob_encode_string(const char *s, char *buf, int64_t cap) { return copy_out(gob::encode_string(s), buf, cap); }
int32_t gh_gob_decode_int(const char *buf, int64_t n, int64_t *out) {
    return guard([&] { *out = gob::decode_int(std::string(buf, (size_t)n)); });
}
// kinds: 0 int, 1 float64, 2 bool
int64_t gh_gob_encode_params(const char **names, const double *vals, const int32_t *kinds, int32_t n, char *buf, int64_t cap) {
    gob::Entries e;
    for (int32_t k = 0; k < n; k++) {
        gob::Value v = kinds[k] == 0 ? gob::Value::of_int((int64_t)vals[k]) : gob::Value::of_float(vals[k]);
        if (kinds[k] == 2) v.kind = gob::Value::Bool, v.b = vals[k] != 0;
        e.emplace_back(names[k], v);
    }
    return copy_out(gob::encode_map("Params", e), buf, cap);
}
// decoded entries as "name\tkind\tvalue" lines
int64_t gh_gob_decode_params(const char *buf, int64_t n, char *out, int64_t cap) {
    std::string text;
    int32_t rc = guard([&] {
        for (auto &kv : gob::decode_map(std::string(buf, (size_t)n))) {
            char num[64];
            snprintf(num, sizeof(num), "%.17g", kv.second.number());
            text += kv.first + "\t" + std::to_string((int)kv.second.kind) + "\t" +
                    (kv.second.kind == gob::Value::String ? kv.second.s : std::string(num)) + "\n";
        }
    });
    return rc != 0 ? rc : copy_out(text, out, cap);
}
// the worked example of the encoding/gob documentation, type Point struct{ X, Y int } holding {22, 33}, rebuilt from the
// primitives: the definition of type 65 (a structType with two int fields), then the value
int64_t gh_gob_doc_example(char *buf, int64_t cap) {
    std::string def;
    gob::put_int(def, -65);
    gob::put_uint(def, 3);  // wireType.StructT
    gob::put_uint(def, 1);  // CommonType
    gob::put_uint(def, 1);
    gob::put_string(def, "Point");
    gob::put_uint(def, 1);
    gob::put_int(def, 65);
    gob::put_uint(def, 0);
    gob::put_uint(def, 1);  // Field
    gob::put_uint(def, 2);  // two fieldTypes
    for (const char *f : {"X", "Y"}) {
        gob::put_uint(def, 1);
        gob::put_string(def, f);
        gob::put_uint(def, 1);
        gob::put_int(def, gob::tInt);
        gob::put_uint(def, 0);
    }
    gob::put_uint(def, 0);
    gob::put_uint(def, 0);
    std::string val;
    gob::put_int(val, 65);
    gob::put_uint(val, 1);
    gob::put_int(val, 22);
    gob::put_uint(val, 1);
    gob::put_int(val, 33);
    gob::put_uint(val, 0);
    return copy_out(gob::message(def) + gob::message(val), buf, cap);
}
void *gh_mfusers_new() { return new logics::MatrixFactorizationUsers(); }
void gh_mfusers_free(void *u) { delete (logics::MatrixFactorizationUsers *)u; }
void gh_mfusers_add(void *u, const char *id, const float *v, int32_t d) {
    ((logics::MatrixFactorizationUsers *)u)->Add(id, std::vector<float>(v, v + d));
}
int32_t gh_mfusers_get(void *u, const char *id, float *out, int32_t cap) {
    std::vector<float> v;
    if (!((logics::MatrixFactorizationUsers *)u)->Get(id, v)) return -1;
    for (size_t t = 0; t < v.size() && (int32_t)t < cap; t++) out[t] = v[t];
    return (int32_t)v.size();
}
int64_t gh_mfusers_marshal(void *u, char *buf, int64_t cap) {
    return copy_out(((logics::MatrixFactorizationUsers *)u)->Marshal(), buf, cap);
}
int32_t gh_mfusers_unmarshal(void *u, const char *buf, int64_t n) {
    return guard([&] { ((logics::MatrixFactorizationUsers *)u)->Unmarshal(std::string(buf, (size_t)n)); });
}

// ---- heap ----------------------------------------------------------------------------------------
int32_t gh_topk_filter(int32_t k, const int32_t *items, const float *weights, int32_t n, int32_t *out_items, float *out_w) {
    heap::TopKFilter f(k);
    for (int32_t t = 0; t < n; t++) f.Push(items[t], weights[t]);
    auto r = f.PopAll();
    for (size_t t = 0; t < r.size(); t++) out_items[t] = r[t].Value, out_w[t] = r[t].Weight;
    return (int32_t)r.size();
}
int32_t gh_pq_drain(int32_t desc, int32_t reverse, const int32_t *items, const float *weights, int32_t n, int32_t *out_items,
                    float *out_w) {
    int32_t cnt = 0;
    int32_t rc = guard([&] {
        heap::PriorityQueue pq(desc != 0);
        for (int32_t t = 0; t < n; t++) pq.Push(items[t], weights[t]);
        heap::PriorityQueue q = reverse ? pq.Reverse() : pq.Clone();
        while (q.Len() > 0) {
            auto e = q.Pop();
            out_items[cnt] = e.first, out_w[cnt] = e.second, cnt++;
        }
    });
    return rc < 0 ? rc : cnt;
}

// ---- ann.Bruteforce ---------------------------------------------------------------------------------
void *gh_bruteforce_new(int32_t metric) { return new ann::Bruteforce(metric); }
void gh_bruteforce_free(void *b) { delete (ann::Bruteforce *)b; }
int32_t gh_bruteforce_add(void *b, const float *v, int32_t d, int32_t *ret) {
    return guard([&] { *ret = ((ann::Bruteforce *)b)->Add(std::vector<float>(v, v + d)); });
}
int32_t gh_bruteforce_search_index(void *b, int32_t q, int32_t k, int32_t prune0, int32_t *idx, float *dist, int32_t *cnt) {
    return guard([&] {
        auto r = ((ann::Bruteforce *)b)->SearchIndex(q, k, prune0 != 0);
        *cnt = (int32_t)r.size();
        for (size_t t = 0; t < r.size(); t++) idx[t] = r[t].first, dist[t] = r[t].second;
    });
}
int32_t gh_bruteforce_search_vector(void *b, const float *q, int32_t d, int32_t k, int32_t prune0, int32_t *idx, float *dist,
                                    int32_t *cnt) {
    return guard([&] {
        auto r = ((ann::Bruteforce *)b)->SearchVector(std::vector<float>(q, q + d), k, prune0 != 0);
        *cnt = (int32_t)r.size();
        for (size_t t = 0; t < r.size(); t++) idx[t] = r[t].first, dist[t] = r[t].second;
    });
}

// ---- vectors.Database ("hip://") ----------------------------------------------------------------------
// Vectors cross the boundary one field at a time through a thread-local staging list: *_stage_* fill it before
// AddVectors, GetVectors / QueryVectors leave their results in it for the gh_vdb_result_* readers.
namespace {
typedef int32_t (*gh_search_cb)(const float *X, int64_t n, int32_t d, int32_t metric, const float *Q, int64_t nq, int32_t k,
                                int32_t *idx, float *dist, int32_t *cnt);
struct CallbackSearcher : vectors::Searcher {  // the CPU test-suite's checker (an exact search built on the oracle)
    gh_search_cb cb;
    explicit CallbackSearcher(gh_search_cb c) : cb(c) {}
    void invalidate(const std::string &) override {}
    void search(const std::string &, const float *X, int64_t n, int d, int metric, const float *Q, int64_t nq, int k,
                int32_t *idx, float *dist, int32_t *cnt) override {
        if (cb(X, n, d, metric, Q, nq, k, idx, dist, cnt) != 0) throw std::runtime_error("search callback failed");
    }
};
typedef int32_t (*gh_sparse_search_cb)(int64_t n, const int64_t *indptr, const uint32_t *indices, const float *values,
                                       const uint8_t *admissible, int64_t nq, const int64_t *q_indptr, const uint32_t *q_indices,
                                       const float *q_values, int32_t k, int32_t *idx, float *score, int32_t *cnt);
struct CallbackSparseSearcher : vectors::SparseSearcher {  // the CPU test-suite's checker for sparse collections
    gh_sparse_search_cb cb;
    explicit CallbackSparseSearcher(gh_sparse_search_cb c) : cb(c) {}
    void invalidate(const std::string &) override {}
    void search(const std::string &, int64_t n, const int64_t *indptr, const uint32_t *indices, const float *values,
                const uint8_t *admissible, int64_t nq, const int64_t *q_indptr, const uint32_t *q_indices, const float *q_values,
                int k, int32_t *idx, float *score, int32_t *cnt) override {
        if (cb(n, indptr, indices, values, admissible, nq, q_indptr, q_indices, q_values, k, idx, score, cnt) != 0)
            throw std::runtime_error("sparse search callback failed");
    }
};
struct VdbHandle {
    std::shared_ptr<vectors::HipDatabase> db;
};
thread_local std::vector<vectors::ScoredVector> g_stage;
thread_local std::vector<int64_t> g_stage_split;  // batch queries: result t is g_stage[split[t] .. split[t+1])
std::vector<std::string> split_lines(const char *s) {
    std::vector<std::string> out;
    if (!s || !*s) return out;
    std::string cur;
    for (const char *p = s;; p++) {
        if (*p == '\n' || *p == 0) {
            out.push_back(cur);
            cur.clear();
            if (*p == 0) break;
        } else {
            cur.push_back(*p);
        }
    }
    return out;
}
vectors::HipDatabase &vdb(void *h) { return *((VdbHandle *)h)->db; }
}  // namespace

// MatrixFactorizationItems (logics/cf.go:36-128); cb = a search callback (the CPU test-suite's checker) or NULL for the GPU
void *gh_mfitems_new(int64_t timestamp_unix_nanos, gh_search_cb cb) {
    std::shared_ptr<vectors::Searcher> s;
    if (cb) s = std::make_shared<CallbackSearcher>(cb);
    return new logics::MatrixFactorizationItems(timestamp_unix_nanos, s);
}
void gh_mfitems_free(void *m) { delete (logics::MatrixFactorizationItems *)m; }
void gh_mfitems_add(void *m, const char *id, const float *v, int32_t d) {
    ((logics::MatrixFactorizationItems *)m)->Add(id, std::vector<float>(v, v + d));
}
int64_t gh_mfitems_count(void *m) { return (int64_t)((logics::MatrixFactorizationItems *)m)->Count(); }
int32_t gh_mfitems_dimension(void *m) { return ((logics::MatrixFactorizationItems *)m)->Dimension(); }
int64_t gh_mfitems_timestamp(void *m) { return ((logics::MatrixFactorizationItems *)m)->Timestamp(); }
int64_t gh_mfitems_id(void *m, int64_t i, char *buf, int64_t cap) {
    return copy_out(((logics::MatrixFactorizationItems *)m)->Id((size_t)i), buf, cap);
}
void gh_mfitems_row(void *m, int64_t i, float *out) {
    auto *x = (logics::MatrixFactorizationItems *)m;
    std::memcpy(out, x->Row((size_t)i), (size_t)x->Dimension() * sizeof(float));
}
int64_t gh_mfitems_marshal(void *m, char *buf, int64_t cap) {
    return copy_out(((logics::MatrixFactorizationItems *)m)->Marshal(), buf, cap);
}
// the blob with the reference's own index section and a device-built graph (MarshalReference); called twice by the ctypes wrapper
// (size, then content): the bytes of the first call are kept until the second has copied them
int64_t gh_mfitems_marshal_reference(void *m, char *buf, int64_t cap) {
    // kept for: the object AND the state it was marshalled from (items added between the two calls make a fresh blob)
    static thread_local std::string kept;
    static thread_local void *kept_for = nullptr;
    static thread_local size_t kept_count = 0;
    static thread_local int64_t kept_stamp = 0;
    int64_t out = -1;
    guard([&] {
        auto *items = (logics::MatrixFactorizationItems *)m;
        if (!(buf && kept_for == m && kept_count == items->Count() && kept_stamp == items->Timestamp() && (int64_t)kept.size() <= cap)) {
            kept = items->MarshalReference();
            kept_for = m;
            kept_count = items->Count();
            kept_stamp = items->Timestamp();
        }
        out = copy_out(kept, buf, cap);
        if (buf) {
            kept.clear();
            kept_for = nullptr;
        }
    });
    return out;
}
int32_t gh_mfitems_unmarshal(void *m, const char *buf, int64_t n) {
    return guard([&] { ((logics::MatrixFactorizationItems *)m)->Unmarshal(std::string(buf, (size_t)n)); });
}
// Search: ids joined by '\n' into ids_buf (returns the byte count), scores into scores[0..cap_scores); *n_out = results
int64_t gh_mfitems_search(void *m, const float *v, int32_t d, int32_t n, char *ids_buf, int64_t cap, double *scores,
                          int32_t cap_scores, int32_t *n_out) {
    int64_t bytes = -1;
    const int32_t rc = guard([&] {
        auto res = ((logics::MatrixFactorizationItems *)m)->Search(std::vector<float>(v, v + d), n);
        std::string joined;
        for (size_t t = 0; t < res.size(); t++) {
            if (t) joined.push_back('\n');
            joined += res[t].Id;
            if ((int32_t)t < cap_scores) scores[t] = res[t].Value;
        }
        *n_out = (int32_t)res.size();
        bytes = copy_out(joined, ids_buf, cap);
    });
    return rc == 0 ? bytes : -1;
}
int64_t gh_gob_f32_slice(const float *v, int32_t n, char *buf, int64_t cap) { return copy_out(gob::encode_f32_slice(v, (size_t)n), buf, cap); }

void *gh_vdb_open(const char *url) {
    void *out = nullptr;
    guard([&] { out = new VdbHandle{vectors::Open(url)}; });
    return out;
}
void *gh_vdb_open_with_searcher(gh_search_cb cb) {
    return new VdbHandle{std::make_shared<vectors::HipDatabase>(std::make_shared<CallbackSearcher>(cb))};
}
void *gh_vdb_open_with_searchers(gh_search_cb cb, gh_sparse_search_cb scb) {
    std::shared_ptr<vectors::Searcher> dense;
    std::shared_ptr<vectors::SparseSearcher> sparse;
    if (cb) dense = std::make_shared<CallbackSearcher>(cb);
    if (scb) sparse = std::make_shared<CallbackSparseSearcher>(scb);
    return new VdbHandle{std::make_shared<vectors::HipDatabase>(dense, sparse)};
}
void gh_vdb_free(void *h) { delete (VdbHandle *)h; }
int32_t gh_vdb_close(void *h) { return guard([&] { vdb(h).Close(); }); }
int32_t gh_vdb_add_collection(void *h, const char *name, int32_t dim, int32_t distance, const char *qtype, int32_t bits) {
    return guard([&] {
        vectors::VectorConfig cfg;
        cfg.Type = qtype ? qtype : "";
        cfg.Bits = bits;
        vdb(h).AddCollection(name, dim, (vectors::Distance)distance, cfg);
    });
}
int32_t gh_vdb_delete_collection(void *h, const char *name) { return guard([&] { vdb(h).DeleteCollection(name); }); }
int32_t gh_vdb_describe(void *h, const char *name, int32_t *dim, int32_t *distance, int32_t *bits) {
    return guard([&] {
        auto info = vdb(h).DescribeCollection(name);
        *dim = info.Dimension;
        *distance = (int32_t)info.Dist;
        *bits = info.Config.Bits;
    });
}
// '\n'-joined names into buf (NUL-terminated); returns the length needed or a negative error
int64_t gh_vdb_list(void *h, char *buf, int64_t cap) {
    std::string joined;
    int32_t rc = guard([&] {
        for (auto &n : vdb(h).ListCollections()) joined += (joined.empty() ? "" : "\n") + n;
    });
    if (rc != 0) return rc;
    if ((int64_t)joined.size() + 1 <= cap) std::memcpy(buf, joined.c_str(), joined.size() + 1);
    return (int64_t)joined.size() + 1;
}
int64_t gh_vdb_count(void *h, const char *name) {
    int64_t n = 0;
    int32_t rc = guard([&] { n = vdb(h).CountVectors(name); });
    return rc != 0 ? rc : n;
}
void gh_vdb_stage_clear() {
    g_stage.clear();
    g_stage_split.clear();
}
void gh_vdb_stage_vector(const char *id, const float *values, int32_t n_values, const uint32_t *indices, int32_t n_indices,
                         int32_t hidden, const char *categories, int64_t timestamp_ms) {
    vectors::ScoredVector v;
    v.Id = id;
    v.Values.assign(values, values + n_values);
    if (n_indices > 0) v.Indices.assign(indices, indices + n_indices);
    v.IsHidden = hidden != 0;
    v.Categories = split_lines(categories);
    v.TimestampMs = timestamp_ms;
    g_stage.push_back(std::move(v));
}
int32_t gh_vdb_add_staged(void *h, const char *name) {
    return guard([&] {
        std::vector<vectors::Vector> vs(g_stage.begin(), g_stage.end());
        g_stage.clear();
        vdb(h).AddVectors(name, vs);
    });
}
int32_t gh_vdb_get(void *h, const char *name, const char *ids) {
    return guard([&] {
        gh_vdb_stage_clear();
        for (auto &v : vdb(h).GetVectors(name, split_lines(ids))) {
            vectors::ScoredVector s;
            static_cast<vectors::Vector &>(s) = v;
            g_stage.push_back(std::move(s));
        }
    });
}
int32_t gh_vdb_delete_vectors(void *h, const char *name, int64_t timestamp_ms) {
    return guard([&] { vdb(h).DeleteVectors(name, timestamp_ms); });
}
// the query is the LAST staged vector (dense Values, or Indices + Values for a sparse collection); results replace the staging list
int32_t gh_vdb_query_staged(void *h, const char *name, const char *categories, int32_t topk) {
    return guard([&] {
        if (g_stage.empty()) throw std::invalid_argument("no staged query vector");
        vectors::Vector q = g_stage.back();
        gh_vdb_stage_clear();
        g_stage = vdb(h).QueryVectors(name, q, split_lines(categories), topk);
    });
}
int32_t gh_vdb_query_batch(void *h, const char *name, const float *Q, int64_t nq, int32_t d, const char *categories,
                           int32_t topk) {
    return guard([&] {
        gh_vdb_stage_clear();
        auto res = vdb(h).QueryVectorsBatch(name, std::vector<float>(Q, Q + nq * d), nq, split_lines(categories), topk);
        g_stage_split.push_back(0);
        for (auto &r : res) {
            for (auto &v : r) g_stage.push_back(std::move(v));
            g_stage_split.push_back((int64_t)g_stage.size());
        }
    });
}
// every staged vector is one sparse query; results replace the staging list, split per query
int32_t gh_vdb_query_sparse_staged(void *h, const char *name, const char *categories, int32_t topk) {
    return guard([&] {
        std::vector<vectors::Vector> qs(g_stage.begin(), g_stage.end());
        gh_vdb_stage_clear();
        auto res = vdb(h).QuerySparseBatch(name, qs, split_lines(categories), topk);
        g_stage_split.push_back(0);
        for (auto &r : res) {
            for (auto &v : r) g_stage.push_back(std::move(v));
            g_stage_split.push_back((int64_t)g_stage.size());
        }
    });
}
int32_t gh_vdb_result_nnz(int64_t r) { return (int32_t)g_stage[(size_t)r].Indices.size(); }
void gh_vdb_result_indices(int64_t r, uint32_t *out) {
    std::copy(g_stage[(size_t)r].Indices.begin(), g_stage[(size_t)r].Indices.end(), out);
}
int64_t gh_vdb_result_count() { return (int64_t)g_stage.size(); }
int64_t gh_vdb_result_split(int64_t t) { return t >= 0 && t < (int64_t)g_stage_split.size() ? g_stage_split[(size_t)t] : -1; }
const char *gh_vdb_result_id(int64_t r) { return g_stage[(size_t)r].Id.c_str(); }
float gh_vdb_result_score(int64_t r) { return g_stage[(size_t)r].Score; }
int32_t gh_vdb_result_hidden(int64_t r) { return g_stage[(size_t)r].IsHidden ? 1 : 0; }
int64_t gh_vdb_result_timestamp(int64_t r) { return g_stage[(size_t)r].TimestampMs; }
int32_t gh_vdb_result_dim(int64_t r) { return (int32_t)g_stage[(size_t)r].Values.size(); }
void gh_vdb_result_values(int64_t r, float *out) {
    std::copy(g_stage[(size_t)r].Values.begin(), g_stage[(size_t)r].Values.end(), out);
}
int64_t gh_vdb_result_categories(int64_t r, char *buf, int64_t cap) {
    std::string joined;
    for (auto &c : g_stage[(size_t)r].Categories) joined += (joined.empty() ? "" : "\n") + c;
    if ((int64_t)joined.size() + 1 <= cap) std::memcpy(buf, joined.c_str(), joined.size() + 1);
    return (int64_t)joined.size() + 1;
}

// ---- logics: embedding item-to-item / user-to-user over a vectors.Database ----------------------------------
void *gh_vwriter_new(void *h, const char *collection, int32_t distance, int64_t timestamp_ms, int32_t batch) {
    return new logics::VectorWriter(((VdbHandle *)h)->db, collection, (vectors::Distance)distance, timestamp_ms, batch);
}
void *gh_vwriter_new_sparse(void *h, const char *collection, int64_t timestamp_ms, int32_t batch) {  // distance Dot
    return new logics::VectorWriter(((VdbHandle *)h)->db, collection, vectors::Dot, timestamp_ms, batch, true);
}
void gh_vwriter_free(void *w) { delete (logics::VectorWriter *)w; }
// stage the vector a sparse kind writes for one item / user: kind 0 = tags, 1 = feedback (users / items), 2 = auto
void gh_logics_stage_kind_vector(int32_t kind, const char *id, int32_t hidden, const char *categories, int64_t timestamp_ms,
                                 const int32_t *tags, int32_t n_tags, const float *tags_idf, int32_t n_tags_idf,
                                 const int32_t *feedback, int32_t n_feedback, const float *fb_idf, int32_t n_fb_idf) {
    vectors::Vector meta;
    meta.Id = id;
    meta.IsHidden = hidden != 0;
    meta.Categories = split_lines(categories);
    meta.TimestampMs = timestamp_ms;
    std::vector<int32_t> t(tags, tags + n_tags), f(feedback, feedback + n_feedback);
    std::vector<float> ti(tags_idf, tags_idf + n_tags_idf), fi(fb_idf, fb_idf + n_fb_idf);
    vectors::ScoredVector v;
    static_cast<vectors::Vector &>(v) = kind == 0   ? logics::tagsVector(meta, t, ti)
                                        : kind == 1 ? logics::feedbackVector(meta, f, fi)
                                                    : logics::autoVector(meta, t, ti, f, fi);
    g_stage.push_back(std::move(v));
}
// the vector to add is the LAST staged one
int32_t gh_vwriter_add_staged(void *w) {
    return guard([&] {
        if (g_stage.empty()) throw std::invalid_argument("no staged vector");
        vectors::Vector v = g_stage.back();
        gh_vdb_stage_clear();
        ((logics::VectorWriter *)w)->Add(v);
    });
}
int32_t gh_vwriter_clean(void *w) { return guard([&] { ((logics::VectorWriter *)w)->Clean(); }); }
// results: staged as ScoredVector with Score = the cache.Score value (double narrowed for transport is NOT wanted: see
// gh_logics_result_score)
namespace {
thread_local std::vector<double> g_scores;
void stage_scores(const std::vector<std::vector<logics::Score>> &res) {
    gh_vdb_stage_clear();
    g_scores.clear();
    g_stage_split.push_back(0);
    for (auto &r : res) {
        for (auto &sc : r) {
            vectors::ScoredVector v;
            v.Id = sc.Id;
            v.Categories = sc.Categories;
            g_stage.push_back(std::move(v));
            g_scores.push_back(sc.Value);
        }
        g_stage_split.push_back((int64_t)g_stage.size());
    }
}
}  // namespace
int32_t gh_logics_query_similar(void *h, const char *collection, const char *id, const char *categories, int32_t n) {
    return guard([&] { stage_scores({logics::QuerySimilar(vdb(h), collection, id, split_lines(categories), n)}); });
}
int32_t gh_logics_query_similar_bulk(void *h, const char *collection, const char *ids, const char *categories, int32_t n) {
    return guard([&] { stage_scores(logics::QuerySimilarBulk(vdb(h), collection, split_lines(ids), split_lines(categories), n)); });
}
int32_t gh_logics_query_similar_typed(void *h, const char *collection, const char *type, const char *id, const char *categories,
                                      int32_t n) {
    return guard([&] { stage_scores({logics::QuerySimilarTyped(vdb(h), collection, type, id, split_lines(categories), n)}); });
}
int32_t gh_logics_query_similar_typed_bulk(void *h, const char *collection, const char *type, const char *ids,
                                           const char *categories, int32_t n) {
    return guard([&] {
        stage_scores(logics::QuerySimilarTypedBulk(vdb(h), collection, type, split_lines(ids), split_lines(categories), n));
    });
}
double gh_logics_result_score(int64_t r) { return g_scores[(size_t)r]; }

// users: Q is n_users x d; excludes: one '\n'-joined id list per user, users separated by '\x1e' (record separator)
int32_t gh_logics_cf_recommend_bulk(void *h, const char *collection, const float *Q, int64_t n_users, int32_t d,
                                    const char *excludes, int32_t cache_size) {
    return guard([&] {
        std::vector<logics::UserQuery> users((size_t)n_users);
        std::vector<std::string> per_user;
        std::string cur;
        for (const char *p = excludes ? excludes : "";; p++) {
            if (*p == '\x1e' || *p == 0) {
                per_user.push_back(cur);
                cur.clear();
                if (*p == 0) break;
            } else {
                cur.push_back(*p);
            }
        }
        for (int64_t t = 0; t < n_users; t++) {
            users[(size_t)t].Embedding.assign(Q + t * d, Q + (t + 1) * d);
            if ((size_t)t < per_user.size()) users[(size_t)t].Exclude = split_lines(per_user[(size_t)t].c_str());
        }
        stage_scores(logics::CollaborativeRecommendBulk(vdb(h), collection, users, cache_size));
    });
}
// master/tasks.go:925-969: items of model `m` into collaborative_filtering_<model_id> of database `h`, users into a new
// MatrixFactorizationUsers (returned; free with gh_mfusers_free).  hidden: one byte per item or NULL; categories: one
// '\n'-joined list per item, items separated by '\x1e', or NULL
void *gh_publish_cf(void *m, void *h, int64_t model_id, const uint8_t *hidden, int32_t n_hidden, const char *categories,
                    int32_t batch) {
    void *out = nullptr;
    guard([&] {
        std::vector<bool> hid(hidden, hidden + (hidden ? n_hidden : 0));
        std::vector<std::vector<std::string>> cats;
        if (categories) {
            std::string cur;
            for (const char *p = categories;; p++) {
                if (*p == '\x1e' || *p == 0) {
                    cats.push_back(split_lines(cur.c_str()));
                    cur.clear();
                    if (*p == 0) break;
                } else {
                    cur.push_back(*p);
                }
            }
        }
        auto users = logics::PublishCollaborativeFiltering(*(cf::MatrixFactorization *)m, vdb(h), model_id, hid, cats, batch > 0 ? batch : 1024);
        out = new logics::MatrixFactorizationUsers(std::move(users));
    });
    return out;
}
int32_t gh_mfusers_count(void *u) { return (int32_t)((logics::MatrixFactorizationUsers *)u)->Count(); }

// CPU test hook for the ranking kernels' integer encodings (csrc/rank_keys.hpp, the very header the kernels include):
// what = 0 fkey, 1 fkey_inv(fkey), 2 dist_key, 3 dist_from_key(dist_key, -0 flag), 4 score_ord, 5 key_score(make_key(score_ord, row))
// -> out_u32[i]; 6 key_row(make_key(.., row)) with row = i.  Float results come back as their bits.
void gh_test_rank_key(int32_t what, const float *x, int64_t n, uint32_t *out_u32) {
    namespace rk = gorse::rank;
    for (int64_t i = 0; i < n; i++) {
        switch (what) {
        case 0: out_u32[i] = rk::fkey(x[i]); break;
        case 1: out_u32[i] = rk::f2u(rk::fkey_inv(rk::fkey(x[i]))); break;
        case 2: out_u32[i] = rk::dist_key(x[i]); break;
        case 3: out_u32[i] = rk::f2u(rk::dist_from_key(rk::dist_key(x[i]), rk::dist_is_negative_zero(x[i]))); break;
        case 4: out_u32[i] = rk::score_ord(x[i]); break;
        case 5: out_u32[i] = rk::f2u(rk::key_score(rk::make_key(rk::score_ord(x[i]), (int32_t)i))); break;
        default: out_u32[i] = (uint32_t)rk::key_row(rk::make_key(rk::score_ord(x[i]), (int32_t)i)); break;
        }
    }
}
// CPU test hook for the symmetric all-pairs sweep's tile schedule (csrc/topk_sym.hpp, the very header the kernel includes):
// cover[q * n + r] += 1 for every (query q, row r) pair the workgroups of an all-pairs sweep over n rows (queries = the rows
// q0 .. q0 + nq) score -- by a workgroup's own columns (every tile it multiplies) or along the rows of a transposed tile, for the
// columns that emit.  Returns the number of row tiles all workgroups multiplied.
int64_t gh_test_topk_sym_cover(int64_t n, int64_t q0, int64_t nq, int32_t tile_rows, int32_t bq, int32_t *cover) {
    const int64_t all_tiles = (n + tile_rows - 1) / tile_rows;
    const int64_t blocks = (nq + bq - 1) / bq;
    int64_t multiplied = 0;
    for (int64_t blk = 0; blk < blocks; blk++) {
        const gorse::SymSchedule s(q0, nq, tile_rows, bq, blk);
        const int64_t nt = s.tiles(all_tiles);
        multiplied += nt;
        for (int64_t tl = 0; tl < nt; tl++) {
            const int64_t tile = s.tile_index(tl);
            for (int64_t r = tile * tile_rows; r < (tile + 1) * tile_rows && r < n; r++)
                for (int64_t c = blk * bq; c < (blk + 1) * bq; c++) {
                    if (c < nq) cover[c * n + r] += 1;  // the column's own list
                    if (s.transposed(tile) && gorse::SymSchedule::column_emits(q0, nq, tile_rows, c))
                        cover[(r - q0) * n + (q0 + c)] += 1;  // the row's query, candidate = the column's row
                }
        }
    }
    return multiplied;
}
// CPU test hooks for the BPR chunk preparation by user bins (csrc/bpr_bins.hpp, the very header the launch code includes):
// the geometry of a chunk -> out[0..4] = shift, bins, tile, tiles, ok; the words of the tile x bin matrix a handle allocates;
// and the four passes restated on the CPU over given user keys -> run offsets bucket[U + 2] and the (sample id, user) pairs in run order.
void gh_test_bpr_bins_geometry(int64_t U, int64_t n, int64_t *out5) {
    const gorse::PrepBins pb = gorse::prep_bins(U, n);
    out5[0] = pb.shift, out5[1] = pb.nbins, out5[2] = pb.tile, out5[3] = (n + pb.tile - 1) / pb.tile, out5[4] = pb.ok ? 1 : 0;
}
int64_t gh_test_bpr_bins_matrix_words(int64_t U, int64_t cap) { return (int64_t)gorse::prep_matrix_words(U, cap); }
int32_t gh_test_bpr_bins_emulate(int64_t U, const int32_t *key, int64_t n, int32_t *bucket, int32_t *pair_s, int32_t *pair_u) {
    std::vector<int32_t> b, ps, pu;
    if (!gorse::prep_bins_emulate(U, key, n, b, ps, pu)) return 0;
    std::copy(b.begin(), b.end(), bucket);
    std::copy(ps.begin(), ps.end(), pair_s);
    std::copy(pu.begin(), pu.end(), pair_u);
    return 1;
}
// the ALS row plan's long-row threshold for a side of `side_entries` feedbacks (csrc/als_plan.hpp, the header als_build_plan includes)
int64_t gh_test_als_long_row(int64_t side_entries, int32_t d) { return gorse::als_long_row_threshold(side_entries, d); }
// the 64-bit sparse ranking key itself, and the number of results the reference returns (xvec.go:379-446)
uint64_t gh_test_sparse_key(float score, int32_t row) { return gorse::rank::make_key(gorse::rank::score_ord(score), row); }
int32_t gh_test_sparse_written(int64_t pos, int64_t neg, int64_t adm, int32_t k) { return gorse::rank::written(pos, neg, adm, k); }
// the scratch numbering of a sparse index's rows (csrc/sparse_host.hpp, the header gorse_sparse_create includes): longest first, the
// rows longer than front_cut in a row group of their own with phantom ids behind them.  new_of: n, orig_of: n + group entries at most;
// out3 = {scratch ids, front rows, phantom ids}
void gh_test_sparse_row_order(int64_t n, const int64_t *indptr, int64_t front_cut, int64_t group, int32_t *new_of, int32_t *orig_of,
                              int64_t *out3) {
    const gorse::sparse::RowOrder o = gorse::sparse::order_rows(n, indptr, front_cut, group);
    for (int64_t r = 0; r < n; r++) new_of[r] = o.new_of[(size_t)r];
    for (int64_t s = 0; s < o.Np; s++) orig_of[s] = o.orig_of[(size_t)s];
    out3[0] = o.Np, out3[1] = o.n_front, out3[2] = o.pad;
}
// the HNSW levels MarshalReference gives the n vectors of a model (gorse_vectors.hpp ReferenceLevels: the arithmetic both twins share)
// and the levelFactor it writes, as its bits
uint32_t gh_test_hnsw_levels(int64_t n, int32_t *level) {
    const std::vector<int> lv = logics::MatrixFactorizationItems::ReferenceLevels(n);
    for (int64_t i = 0; i < n; i++) level[i] = lv[(size_t)i];
    const float f = logics::MatrixFactorizationItems::ReferenceLevelFactor();
    uint32_t bits;
    memcpy(&bits, &f, 4);
    return bits;
}
}  // extern "C"
