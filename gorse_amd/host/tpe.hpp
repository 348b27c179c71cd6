// tpe.hpp -- the tree-structured Parzen estimator sampler that master/tasks.go:1297-1300 gives its study
// (goptuna.StudyOptionSampler(tpe.NewSampler())).  The sampler lives in github.com/c-bata/goptuna v0.9.0 (go.mod), which is
// absent from /root/reference; goptuna documents it as a port of Optuna's TPESampler, whose algorithm is published (Bergstra
// et al., "Algorithms for Hyper-Parameter Optimization", NIPS 2011; Optuna's _tpe/sampler.py and parzen_estimator.py).  What
// is restated here is that algorithm with the defaults both implementations share:
//   * the first 10 trials of a parameter are drawn at random (n_startup_trials);
//   * afterwards the finished trials are split by objective value into the best gamma(n) = min(ceil(0.1 n), 25) ("below")
//     and the rest ("above");
//   * each side becomes a Parzen estimator over the parameter's (transformed) range: one Gaussian per observation plus a
//     prior Gaussian at the middle of the range with sigma = the range; an observation's sigma is the larger gap to its
//     neighbours (the range's ends count as neighbours of the outermost observations only when consider_endpoints, which is
//     off), clipped to [range / min(100, 1 + n), range] ("magic clip"); weights: 1 per observation while n < 25, else a ramp
//     from 1 / n to 1 over the oldest n - 25 and 1 for the newest 25; prior weight 1;
//   * 24 candidates (n_ei_candidates) are drawn from the "below" mixture truncated to the range, and the one with the largest
//     log l(x) - log g(x) is suggested;
//   * log-uniform parameters are handled in log space, discrete ones (step q) on [low - q/2, high + q/2] with the mass of a
//     candidate's bucket instead of its density, categorical ones by weighted counts + prior.
// PARITY UNPINNED: no test of the reference fixes a TPE draw (optimize_test.go's TestTPE only needs the best of a 4 x 4 grid
// to be found in 10 trials), goptuna's source is not here, and Go's math/rand stream is not reproducible without Go.
#pragma once
#include <algorithm>
#include <cmath>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace gorse {
namespace cf {
namespace tpe {

constexpr int kStartupTrials = 10, kEiCandidates = 24;
constexpr double kPriorWeight = 1.0, kEps = 1e-12;

inline int default_gamma(int n) { return std::min((int)std::ceil(0.1 * n), 25); }
inline std::vector<double> default_weights(int n) {
    std::vector<double> w((size_t)std::max(n, 0), 1.0);
    if (n >= 25) {
        const int ramp = n - 25;
        for (int i = 0; i < ramp; i++) w[(size_t)i] = ramp == 1 ? 1.0 / n : 1.0 / n + (1.0 - 1.0 / n) * i / (ramp - 1);  // linspace(1/n, 1, n - 25)
    }
    return w;
}
inline double normal_cdf(double x, double mu, double sigma) { return 0.5 * (1.0 + std::erf((x - mu) / (sigma * 1.4142135623730951))); }

// observations in the order the trials finished (the weights favour the newest)
struct ParzenEstimator {
    std::vector<double> weights, mus, sigmas;
    ParzenEstimator(const std::vector<double> &obs, double low, double high) {
        const double prior_mu = 0.5 * (low + high), prior_sigma = high - low;
        const size_t n = obs.size();
        std::vector<size_t> order(n);
        for (size_t i = 0; i < n; i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return obs[a] < obs[b]; });
        size_t prior_pos = 0;
        while (prior_pos < n && obs[order[prior_pos]] < prior_mu) prior_pos++;  // searchsorted (left)
        const std::vector<double> w_obs = default_weights((int)n);
        for (size_t i = 0; i <= n; i++) {
            if (i == prior_pos) {
                mus.push_back(prior_mu);
                weights.push_back(kPriorWeight);
            }
            if (i < n) {
                mus.push_back(obs[order[i]]);
                weights.push_back(w_obs[order[i]]);
            }
        }
        const size_t m = mus.size();
        sigmas.assign(m, prior_sigma);
        if (n > 0) {
            for (size_t i = 0; i < m; i++) {
                const double left = i == 0 ? low : mus[i - 1], right = i + 1 == m ? high : mus[i + 1];
                sigmas[i] = std::max(mus[i] - left, right - mus[i]);
            }
            if (m >= 2) {  // consider_endpoints = false: the outermost observations look inwards only
                sigmas[0] = mus[1] - mus[0];
                sigmas[m - 1] = mus[m - 1] - mus[m - 2];
            }
        }
        sigmas[prior_pos] = prior_sigma;
        const double maxsigma = high - low, minsigma = (high - low) / std::min(100.0, 1.0 + (double)m);
        double total = 0;
        for (size_t i = 0; i < m; i++) {
            sigmas[i] = std::min(std::max(sigmas[i], minsigma), maxsigma);
            total += weights[i];
        }
        for (double &w : weights) w /= total;
    }
};

// RNG: anything with Float64() in [0, 1) and NormFloat64()
template <class Rng>
std::vector<double> sample_from_gmm(const ParzenEstimator &pe, double low, double high, double q, int n, Rng &rng) {
    std::vector<double> out;
    if (!(low < high)) {  // a range of one point
        out.assign((size_t)n, low);
        return out;
    }
    while ((int)out.size() < n) {
        double u = rng.Float64(), acc = 0;
        size_t c = pe.weights.size() - 1;
        for (size_t i = 0; i < pe.weights.size(); i++) {
            acc += pe.weights[i];
            if (u < acc) {
                c = i;
                break;
            }
        }
        const double x = pe.mus[c] + pe.sigmas[c] * rng.NormFloat64();
        if (x < low || x >= high) continue;  // truncation by rejection
        out.push_back(q > 0 ? std::round(x / q) * q : x);
    }
    return out;
}

inline std::vector<double> gmm_log_pdf(const std::vector<double> &samples, const ParzenEstimator &pe, double low, double high, double q) {
    std::vector<double> out(samples.size(), 0.0);
    if (!(low < high)) return out;
    const size_t m = pe.mus.size();
    double p_accept = 0;
    for (size_t i = 0; i < m; i++)
        p_accept += pe.weights[i] * (normal_cdf(high, pe.mus[i], pe.sigmas[i]) - normal_cdf(low, pe.mus[i], pe.sigmas[i]));
    for (size_t s = 0; s < samples.size(); s++) {
        const double x = samples[s];
        if (q > 0) {  // the mass of the bucket [x - q/2, x + q/2] inside the range
            const double ub = std::min(x + q / 2, high), lb = std::max(x - q / 2, low);
            double prob = 0;
            for (size_t i = 0; i < m; i++) prob += pe.weights[i] * (normal_cdf(ub, pe.mus[i], pe.sigmas[i]) - normal_cdf(lb, pe.mus[i], pe.sigmas[i]));
            out[s] = std::log(prob + kEps) - std::log(p_accept + kEps);
        } else {  // logsumexp over the components
            std::vector<double> t(m);
            double mx = -INFINITY;
            for (size_t i = 0; i < m; i++) {
                const double z = (x - pe.mus[i]) / std::max(pe.sigmas[i], kEps);
                t[i] = -0.5 * z * z - std::log(std::max(pe.sigmas[i], kEps) * 2.5066282746310002) + std::log(pe.weights[i]) - std::log(p_accept);
                mx = std::max(mx, t[i]);
            }
            double sum = 0;
            for (size_t i = 0; i < m; i++) sum += std::exp(t[i] - mx);
            out[s] = mx + std::log(sum);
        }
    }
    return out;
}

// what the study remembers of a finished trial
struct Finished {
    double value;                           // the objective (maximised)
    std::map<std::string, double> params;   // internal representation: the value itself, or the choice's position
};

// the trials that have parameter `name`, split into (below, above) by objective value, each in finishing order
inline void split(const std::vector<Finished> &done, const std::string &name, std::vector<double> &below, std::vector<double> &above) {
    std::vector<size_t> have;
    for (size_t i = 0; i < done.size(); i++)
        if (done[i].params.count(name)) have.push_back(i);
    std::vector<size_t> by_value = have;
    std::stable_sort(by_value.begin(), by_value.end(), [&](size_t a, size_t b) { return done[a].value > done[b].value; });
    const size_t n_below = (size_t)default_gamma((int)have.size());
    std::vector<char> is_below(done.size(), 0);
    for (size_t i = 0; i < n_below && i < by_value.size(); i++) is_below[by_value[i]] = 1;
    for (size_t i : have) (is_below[i] ? below : above).push_back(done[i].params.at(name));
}

template <class Rng>
double suggest_numerical(const std::vector<double> &below, const std::vector<double> &above, double low, double high, double q, Rng &rng) {
    const ParzenEstimator pb(below, low, high), pa(above, low, high);
    const std::vector<double> cand = sample_from_gmm(pb, low, high, q, kEiCandidates, rng);
    const std::vector<double> lb = gmm_log_pdf(cand, pb, low, high, q), la = gmm_log_pdf(cand, pa, low, high, q);
    size_t best = 0;
    for (size_t i = 1; i < cand.size(); i++)
        if (lb[i] - la[i] > lb[best] - la[best]) best = i;
    return cand[best];
}

template <class Rng>
int suggest_categorical(const std::vector<double> &below, const std::vector<double> &above, int n_choices, Rng &rng) {
    auto posterior = [&](const std::vector<double> &obs) {
        std::vector<double> w((size_t)n_choices, kPriorWeight);
        const std::vector<double> ow = default_weights((int)obs.size());
        for (size_t i = 0; i < obs.size(); i++) {
            const int c = (int)obs[i];
            if (c >= 0 && c < n_choices) w[(size_t)c] += ow[i];
        }
        double total = 0;
        for (double x : w) total += x;
        for (double &x : w) x /= total;
        return w;
    };
    const std::vector<double> pb = posterior(below), pa = posterior(above);
    int best = -1;
    double best_score = 0;
    for (int s = 0; s < kEiCandidates; s++) {
        double u = rng.Float64(), acc = 0;
        int c = n_choices - 1;
        for (int i = 0; i < n_choices; i++) {
            acc += pb[(size_t)i];
            if (u < acc) {
                c = i;
                break;
            }
        }
        const double score = std::log(pb[(size_t)c]) - std::log(pa[(size_t)c]);
        if (best < 0 || score > best_score) best = c, best_score = score;
    }
    return best;
}

}  // namespace tpe
}  // namespace cf
}  // namespace gorse
