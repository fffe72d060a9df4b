// ans.cpp -- range-ANS entropy coder for the compression report (host code).
//
// Replaces the coder the reference calls for its "real" bit counts: constriction's AnsCoder with a QuantizedGaussian model
// (lib/entropy_model.py:46-62, consumed as `real_bitrate` in train_nerv_compression.py:484-512, 560-577) and with a Categorical
// model (lib/entropy_model.py:65-81).  constriction (a Rust crate with Python bindings) is not part of the reference tree; this
// file restates its published default configuration:
//   * stream code: rANS with a 64-bit state, 32-bit words, 24-bit fixed-point probabilities; symbols are encoded in REVERSE
//     order (stack semantics) so that decoding pops them in message order; the compressed message is the emitted words followed
//     by the non-zero words of the final state;
//   * QuantizedGaussian(min, max, mean, std): the "leaky" quantisation -- every integer in [min, max] owns at least one of the
//     2^24 slots, the remaining 2^24 - n are split by the Gaussian mass of [k - 1/2, k + 1/2), renormalised to the support;
//   * Categorical(p): fixed-point approximation of p with every symbol >= 1 slot.
// Bit-exact equality with constriction's byte stream is NOT claimed (its tie-breaking in the fixed-point rounding is not
// specified by the algorithm); what is tested is decode(encode(x)) == x and  coded bits <= ideal code length * 1.01 + 64.
#include "common.h"
#include <math.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

namespace {

constexpr int PREC = 24;
constexpr uint32_t TOTAL = 1u << PREC;

struct Table {                       // cumulative slot counts: symbol i owns [cdf[i], cdf[i + 1])
    std::vector<uint32_t> cdf;
    int n() const { return (int)cdf.size() - 1; }
};

inline double phi(double x) { return 0.5 * erfc(-x * 0.70710678118654752440); }

bool gaussian_table(int32_t lo, int32_t hi, double mean, double std, Table& t) {
    if (hi < lo || !(std > 0.0) || !isfinite(mean) || !isfinite(std)) return false;
    const int64_t n = (int64_t)hi - lo + 1;
    if (n > (int64_t)TOTAL / 2) return false;
    const double free_slots = (double)(TOTAL - (uint32_t)n);
    const double c0 = phi((lo - 0.5 - mean) / std), c1 = phi((hi + 0.5 - mean) / std);
    const double z = c1 - c0;
    t.cdf.resize((size_t)n + 1);
    t.cdf[0] = 0;
    for (int64_t k = 1; k < n; ++k) {
        double f = z > 0.0 ? (phi((lo + k - 0.5 - mean) / std) - c0) / z : (double)k / (double)n;
        f = f < 0.0 ? 0.0 : (f > 1.0 ? 1.0 : f);
        t.cdf[(size_t)k] = (uint32_t)(free_slots * f) + (uint32_t)k;             // leak: one slot per symbol below k
    }
    t.cdf[(size_t)n] = TOTAL;
    for (int64_t k = 1; k <= n; ++k)                                               // monotone by construction; keep it strict
        if (t.cdf[(size_t)k] <= t.cdf[(size_t)k - 1]) return false;
    return true;
}

bool categorical_table(const double* p, int K, Table& t) {
    if (K < 1 || K > (int)(TOTAL / 2)) return false;
    double sum = 0.0;
    for (int i = 0; i < K; ++i) { if (!(p[i] >= 0.0) || !isfinite(p[i])) return false; sum += p[i]; }
    if (!(sum > 0.0)) return false;
    std::vector<uint32_t> f((size_t)K);
    const double free_slots = (double)(TOTAL - (uint32_t)K);
    uint64_t used = 0;
    for (int i = 0; i < K; ++i) { f[(size_t)i] = 1u + (uint32_t)(free_slots * (p[i] / sum)); used += f[(size_t)i]; }
    // hand the slots lost to truncation to the most probable symbols (largest first, index order on ties)
    std::vector<int> order((size_t)K);
    for (int i = 0; i < K; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return p[a] > p[b]; });
    for (size_t j = 0; used < TOTAL; j = (j + 1) % (size_t)K) { ++f[(size_t)order[j]]; ++used; }
    t.cdf.resize((size_t)K + 1);
    t.cdf[0] = 0;
    for (int i = 0; i < K; ++i) t.cdf[(size_t)i + 1] = t.cdf[(size_t)i] + f[(size_t)i];
    return t.cdf[(size_t)K] == TOTAL;
}

// symbols are table indices 0 .. n-1
long encode(const Table& t, const int32_t* sym, size_t count, uint32_t* out, size_t cap) {
    std::vector<uint32_t> rev;                                  // words in emission order; the message is this order REVERSED + state
    rev.reserve(count / 3 + 8);
    uint64_t state = 0;
    const int n = t.n();
    for (size_t i = count; i-- > 0;) {
        const int32_t s = sym[i];
        if (s < 0 || s >= n) return -1;
        const uint32_t left = t.cdf[(size_t)s], prob = t.cdf[(size_t)s + 1] - left;
        if ((state >> (64 - PREC)) >= prob) { rev.push_back((uint32_t)state); state >>= 32; }
        state = ((state / prob) << PREC) | (state % prob + left);
    }
    size_t state_words = state == 0 ? 0 : (state >> 32 ? 2 : 1);
    const size_t total = rev.size() + state_words;
    if (out == nullptr || cap < total) return (long)total;
    // layout: [bulk words in REVERSE emission order (= the order the decoder refills in), low state word, high state word]
    size_t w = 0;
    for (size_t i = rev.size(); i-- > 0;) out[w++] = rev[i];
    if (state_words >= 1) out[w++] = (uint32_t)state;
    if (state_words == 2) out[w++] = (uint32_t)(state >> 32);
    return (long)total;
}

int decode(const Table& t, const uint32_t* words, size_t n_words, size_t count, int32_t* sym) {
    // the state is the LAST one or two words; whether it has two is decided exactly as the encoder did: a two-word state has a
    // non-zero high word, and the encoder writes the high word last
    uint64_t state = 0;
    size_t bulk = n_words;
    // candidates are tried from two state words down; a wrong guess cannot pass the end check (all bulk words consumed, state 0)
    auto run = [&](size_t state_words) -> bool {
        if (state_words > n_words) return false;
        bulk = n_words - state_words;
        state = 0;
        if (state_words >= 1) state = words[bulk];
        if (state_words == 2) { if (words[bulk + 1] == 0) return false; state |= (uint64_t)words[bulk + 1] << 32; }
        if (state_words == 1 && state == 0) return false;
        size_t next = 0;
        for (size_t i = 0; i < count; ++i) {
            const uint32_t q = (uint32_t)(state & (TOTAL - 1));
            const size_t s = (size_t)(std::upper_bound(t.cdf.begin(), t.cdf.end(), q) - t.cdf.begin()) - 1;
            const uint32_t left = t.cdf[s], prob = t.cdf[s + 1] - left;
            sym[i] = (int32_t)s;
            state = (state >> PREC) * prob + (q - left);
            if ((state >> 32) == 0 && next < bulk) state = (state << 32) | words[next++];
        }
        return next == bulk && state == 0;
    };
    for (int sw = 2; sw >= 0; --sw)
        if (run((size_t)sw)) return BNERV_OK;
    return bnerv_set_error(BNERV_E_ARG, "ans_decode: the words do not decode to %zu symbols under this model", count);
}

}  // namespace

extern "C" long bnerv_ans_encode_gaussian(const int32_t* symbols, size_t n, int32_t min_sym, int32_t max_sym, double mean, double std_, uint32_t* out, size_t cap_words) {
    Table t;
    if (!symbols && n) { bnerv_set_error(BNERV_E_ARG, "ans_encode_gaussian: null symbols"); return -1; }
    if (!gaussian_table(min_sym, max_sym, mean, std_, t)) { bnerv_set_error(BNERV_E_ARG, "ans_encode_gaussian: bad model (min %d max %d mean %g std %g)", min_sym, max_sym, mean, std_); return -1; }
    std::vector<int32_t> idx(n);
    for (size_t i = 0; i < n; ++i) {
        if (symbols[i] < min_sym || symbols[i] > max_sym) { bnerv_set_error(BNERV_E_ARG, "ans_encode_gaussian: symbol %d outside [%d, %d]", symbols[i], min_sym, max_sym); return -1; }
        idx[i] = symbols[i] - min_sym;
    }
    return encode(t, idx.data(), n, out, cap_words);
}

extern "C" int bnerv_ans_decode_gaussian(const uint32_t* words, size_t n_words, size_t n, int32_t min_sym, int32_t max_sym, double mean, double std_, int32_t* symbols_out) {
    Table t;
    BNERV_REQUIRE((words || !n_words) && (symbols_out || !n), "ans_decode_gaussian: null buffer");
    BNERV_REQUIRE(gaussian_table(min_sym, max_sym, mean, std_, t), "ans_decode_gaussian: bad model");
    const int rc = decode(t, words, n_words, n, symbols_out);
    if (rc != BNERV_OK) return rc;
    for (size_t i = 0; i < n; ++i) symbols_out[i] += min_sym;
    return BNERV_OK;
}

extern "C" long bnerv_ans_encode_categorical(const int32_t* symbols, size_t n, const double* probs, int K, uint32_t* out, size_t cap_words) {
    Table t;
    if ((!symbols && n) || !probs) { bnerv_set_error(BNERV_E_ARG, "ans_encode_categorical: null argument"); return -1; }
    if (!categorical_table(probs, K, t)) { bnerv_set_error(BNERV_E_ARG, "ans_encode_categorical: bad probabilities (K = %d)", K); return -1; }
    const long r = encode(t, symbols, n, out, cap_words);
    if (r < 0) bnerv_set_error(BNERV_E_ARG, "ans_encode_categorical: symbol outside [0, %d)", K);
    return r;
}

extern "C" int bnerv_ans_decode_categorical(const uint32_t* words, size_t n_words, size_t n, const double* probs, int K, int32_t* symbols_out) {
    Table t;
    BNERV_REQUIRE((words || !n_words) && (symbols_out || !n) && probs, "ans_decode_categorical: null argument");
    BNERV_REQUIRE(categorical_table(probs, K, t), "ans_decode_categorical: bad probabilities");
    return decode(t, words, n_words, n, symbols_out);
}
