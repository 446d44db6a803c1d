// Host-side watermark key derivation: MT19937 + forward Fisher-Yates, i.e. what
// torch.randperm does on the CPU generator that GentimeWatermark._split_with_seed
// re-seeds for every context (wmar/watermarking/gentime_watermark.py:161-174).
// The whole key is built ONCE as a bitmap table so that no per-step host work is left.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>
#include <vector>

#include "common.h"

namespace wmar {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

namespace {

struct Mt19937 {
    uint32_t s[624];
    int idx;
    explicit Mt19937(uint32_t seed) {
        s[0] = seed;
        for (int j = 1; j < 624; ++j) s[j] = 1812433253u * (s[j - 1] ^ (s[j - 1] >> 30)) + (uint32_t)j;
        idx = 624;
    }
    void refill() {
        for (int k = 0; k < 624; ++k) {
            uint32_t y = (s[k] & 0x80000000u) | (s[(k + 1) % 624] & 0x7fffffffu);
            uint32_t v = s[(k + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            s[k] = v;
        }
        idx = 0;
    }
    uint32_t next() {
        if (idx >= 624) refill();
        uint32_t y = s[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
};

// First `take` entries of randperm(n): the forward shuffle fixes position i at step i,
// so the walk can stop early -- but the STREAM must still advance n-1 draws when another
// randperm follows on the same generator.
void randperm_prefix(Mt19937& g, int64_t n, int64_t take, std::vector<int32_t>& r, bool drain) {
    r.resize((size_t)std::max<int64_t>(n, 1));
    for (int64_t i = 0; i < n; ++i) r[(size_t)i] = (int32_t)i;
    int64_t stop = drain ? n - 1 : std::min<int64_t>(take, n - 1);
    for (int64_t i = 0; i < stop; ++i) {
        int64_t z = (int64_t)(g.next() % (uint64_t)(n - i));
        std::swap(r[(size_t)i], r[(size_t)(i + z)]);
    }
}

struct Scratch {
    std::vector<int32_t> pa, pd;
};

// Writes the greenlist ids (reference order) into out; returns the count.
int64_t greenlist(const wmar_key_params& k, uint64_t seed, Scratch& sc, int64_t* out) {
    Mt19937 g((uint32_t)(seed & 0xffffffffu));
    const int64_t gsize = (int64_t)((double)k.vocab_size * k.gamma);
    int64_t n = 0;
    if (k.split_strategy == WMAR_SPLIT_RAND) {
        int64_t take = std::min(gsize, k.vocab_size);
        randperm_prefix(g, k.vocab_size, take, sc.pa, false);
        for (int64_t i = 0; i < take; ++i) out[n++] = sc.pa[(size_t)i];
    } else {
        int64_t na = (int64_t)((double)k.n_alive * k.gamma);
        int64_t nd = gsize - na;
        na = std::min(na, k.n_alive);
        if (nd > k.n_dead) nd = k.n_dead;
        if (nd < 0) nd = std::max<int64_t>(0, k.n_dead + nd);
        randperm_prefix(g, k.n_alive, na, sc.pa, true);
        randperm_prefix(g, k.n_dead, nd, sc.pd, false);
        for (int64_t i = 0; i < na; ++i) out[n++] = k.alive_ids[sc.pa[(size_t)i]];
        for (int64_t i = 0; i < nd; ++i) out[n++] = k.dead_ids[sc.pd[(size_t)i]];
    }
    return n;
}

uint64_t context_seed(uint64_t salt, int64_t ctx_sum) {
    unsigned __int128 p = (unsigned __int128)salt * (unsigned __int128)(uint64_t)ctx_sum;
    return (uint64_t)(p % (unsigned __int128)0xffffffffffffffffULL);
}

}  // namespace
}  // namespace wmar

using namespace wmar;

extern "C" {

const char* wmar_last_error(void) { return g_err.c_str(); }
int wmar_version(void) { return 100; }

int64_t wmar_key_row_words(int64_t vocab_size) { return (vocab_size + 31) / 32; }

int64_t wmar_key_table_rows(int32_t seed_strategy, int32_t context_size, int64_t max_token) {
    if (seed_strategy == WMAR_SEED_FIXED || context_size <= 0) return 1;
    return (int64_t)context_size * (max_token - 1) + 1;
}

int64_t wmar_key_greenlist(const wmar_key_params* key, uint64_t seed, int64_t* out_ids_host) {
    if (!key || !out_ids_host) return WMAR_EINVAL;
    Scratch sc;
    return greenlist(*key, seed, sc, out_ids_host);
}

int wmar_gumbel_key_build(uint64_t seed, int64_t vocab_size, float* rs_host, float* log_rs_host, float* score_host) {
    WMAR_REQUIRE(vocab_size > 0, "gumbel_key_build: bad vocab");
    // torch.rand(V, generator=g) on the CPU generator: one 32-bit draw per element, 24 mantissa
    // bits kept (at::uniform_real_distribution<float>); manual_seed keeps the low 32 bits
    Mt19937 g((uint32_t)(seed & 0xffffffffu));
    for (int64_t v = 0; v < vocab_size; ++v) {
        const float rs = (float)(g.next() & 0xffffffu) * 5.9604644775390625e-08f;
        if (rs_host) rs_host[v] = rs;
        if (log_rs_host) log_rs_host[v] = (float)std::log((double)rs);                  // -inf for rs == 0
        if (score_host) score_host[v] = (float)(-std::log((double)(1.0f - rs)));
    }
    return WMAR_OK;
}

int wmar_key_table_build(const wmar_key_params* key, int64_t row0, int64_t n_rows, uint32_t* out_host,
                         int32_t n_threads) {
    WMAR_REQUIRE(key && out_host && n_rows >= 0 && row0 >= 0, "key_table_build: bad arguments");
    WMAR_REQUIRE(key->split_strategy == WMAR_SPLIT_RAND || key->split_strategy == WMAR_SPLIT_STRATIFIED,
                 "key_table_build: split strategy %d unsupported (clustering is out of scope)", key->split_strategy);
    WMAR_REQUIRE(key->vocab_size > 0 && key->vocab_size < (1ll << 31), "key_table_build: bad vocab");
    for (int64_t i = 0; i < key->n_alive; ++i)
        WMAR_REQUIRE(key->alive_ids[i] >= 0 && key->alive_ids[i] < key->vocab_size, "alive id out of range");
    for (int64_t i = 0; i < key->n_dead; ++i)
        WMAR_REQUIRE(key->dead_ids[i] >= 0 && key->dead_ids[i] < key->vocab_size, "dead id out of range");
    const int64_t words = wmar_key_row_words(key->vocab_size);
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, n_rows));
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        Scratch sc;
        std::vector<int64_t> ids((size_t)key->vocab_size + 8);
        for (;;) {
            int64_t r = next.fetch_add(1);
            if (r >= n_rows) break;
            uint64_t seed = key->seed_strategy == WMAR_SEED_FIXED ? 0 : context_seed(key->salt_key, row0 + r);
            int64_t n = greenlist(*key, seed, sc, ids.data());
            uint32_t* row = out_host + r * words;
            memset(row, 0, sizeof(uint32_t) * (size_t)words);
            for (int64_t i = 0; i < n; ++i) row[ids[(size_t)i] >> 5] |= 1u << (ids[(size_t)i] & 31);
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < nt; ++i) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    return WMAR_OK;
}

}  // extern "C"
