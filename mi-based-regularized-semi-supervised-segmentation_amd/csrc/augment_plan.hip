// Host-side planner of the input pipeline: item seeds -> the job table of miseg_augment_slices, natively.
//
// The reference draws every augmentation parameter from Python's ``random`` under FixRandomSeed(seed)
// (contrastyou/augment/sequential_wrapper.py:27-100, whl:deepclustering2/augment/pil_augment.py get_params, torchvision 0.7
// ColorJitter.get_params).  To return the same batch for the same item seed this file restates CPython's generator
// (Modules/_randommodule.c: MT19937 init_by_array seeding, 53-bit random(), getrandbits; Lib/random.py: uniform, randint via
// _randbelow_with_getrandbits, shuffle) and Pillow's Image.rotate matrix set-up (round(cos, 15) etc.) in C++.  The Python
// planner (miseg_amd/slices.py plan_item / encode_jobs) is the readable twin; tests hold the two equal.  A batch of 32
// slices x 2 views plans in tens of microseconds instead of ~2 ms of interpreter time.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "common.h"

namespace {

struct PyRandom {
    uint32_t mt[624];
    int idx;

    void init_genrand(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    void init_by_array(const uint32_t* key, int len) {
        init_genrand(19650218u);
        int i = 1, j = 0;
        for (int k = (624 > len ? 624 : len); k; --k) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            ++i, ++j;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
            if (j >= len) j = 0;
        }
        for (int k = 623; k; --k) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            ++i;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
        }
        mt[0] = 0x80000000u;
    }
    PyRandom() : idx(0) {}
    explicit PyRandom(uint64_t seed) { seed_with(seed); }
    void seed_with(uint64_t seed) {  // random.seed(int): key = 32-bit words of abs(seed), at least one
        uint32_t key[2] = {(uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32)};
        init_by_array(key, key[1] ? 2 : 1);
        drawn = 0;
    }
    // Only a handful of numbers are drawn per stream, so the block "twist" is done lazily, one word per draw: for k < 227 the
    // new word k depends on the OLD words k, k+1 and k+397 only (genrand_uint32's first loop), none of which has been
    // overwritten yet.  Streams here draw < 32 numbers (kMaxDraws).
    static constexpr int kMaxDraws = 200;
    int drawn = 0;
    uint32_t u32() {
        const int k = drawn++;
        if (k >= kMaxDraws) abort();  // unreachable for the recipes above (<= ~20 draws, rejection loops included)
        const uint32_t mag[2] = {0u, 0x9908b0dfu};
        const uint32_t t = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
        uint32_t y = mt[k + 397] ^ (t >> 1) ^ mag[t & 1u];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    double random() {
        const uint32_t a = u32() >> 5, b = u32() >> 6;
        return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
    }
    double uniform(double a, double b) { return a + (b - a) * random(); }
    uint32_t randbelow(uint32_t n) {  // _randbelow_with_getrandbits, n < 2^32
        if (n == 0) return 0;
        int k = 32 - __builtin_clz(n);
        uint32_t r = u32() >> (32 - k);
        while (r >= n) r = u32() >> (32 - k);
        return r;
    }
    int randint(int a, int b) { return a + (int)randbelow((uint32_t)(b - a + 1)); }
};

// Seeding is a 1871-step serial recurrence per stream and a batch needs 5 streams per slice; streams of different slices are
// independent, so they are seeded kLanes at a time in structure-of-arrays form (the lane loop vectorises / pipelines).
constexpr int kLanes = 8;
void seed_batch(const uint64_t* seeds, int n, PyRandom* out) {
    static thread_local uint32_t mt[624][kLanes];
    for (int base = 0; base < n; base += kLanes) {
        const int m = n - base < kLanes ? n - base : kLanes;
        bool wide = false;
        uint32_t key[kLanes];
        for (int l = 0; l < kLanes; ++l) {
            const uint64_t sd = seeds[base + (l < m ? l : 0)];
            key[l] = (uint32_t)sd;
            wide = wide || (sd >> 32) != 0;
        }
        if (wide) {  // two-word keys: rare (seeds >= 2^32), take the scalar path
            for (int l = 0; l < m; ++l) out[base + l].seed_with(seeds[base + l]);
            continue;
        }
        for (int l = 0; l < kLanes; ++l) mt[0][l] = 19650218u;
        for (int i = 1; i < 624; ++i)
            for (int l = 0; l < kLanes; ++l) mt[i][l] = 1812433253u * (mt[i - 1][l] ^ (mt[i - 1][l] >> 30)) + (uint32_t)i;
        int i = 1;
        for (int k = 624; k; --k) {  // one-word key: j stays 0
            const int p = i - 1;
            for (int l = 0; l < kLanes; ++l) mt[i][l] = (mt[i][l] ^ ((mt[p][l] ^ (mt[p][l] >> 30)) * 1664525u)) + key[l];
            if (++i >= 624) {
                for (int l = 0; l < kLanes; ++l) mt[0][l] = mt[623][l];
                i = 1;
            }
        }
        for (int k = 623; k; --k) {
            const int p = i - 1;
            for (int l = 0; l < kLanes; ++l) mt[i][l] = (mt[i][l] ^ ((mt[p][l] ^ (mt[p][l] >> 30)) * 1566083941u)) - (uint32_t)i;
            if (++i >= 624) {
                for (int l = 0; l < kLanes; ++l) mt[0][l] = mt[623][l];
                i = 1;
            }
        }
        for (int l = 0; l < m; ++l) {
            PyRandom& r = out[base + l];
            for (int w = 0; w < 624; ++w) r.mt[w] = mt[w][l];
            r.mt[0] = 0x80000000u;
            r.idx = 624;
            r.drawn = 0;
        }
    }
}

double py_round15(double v) {  // round(v, 15): correctly rounded decimal, as CPython's dtoa-based float.__round__
    char buf[64];
    snprintf(buf, sizeof buf, "%.15f", v);
    return strtod(buf, nullptr);
}

int fix16(double v) {  // libImaging Geometry.c FIX()
    v = v * 65536.0 + 0.5;
    return v < 0.0 ? (int)floor(v) : (int)v;
}

struct Op { int v[9]; };

int rotation_ops(double angle, int w, int h, Op* out) {  // Pillow Image.rotate, NEAREST, expand=False, centre default
    angle = fmod(angle, 360.0);
    if (angle != 0.0 && angle < 0.0) angle += 360.0;      // Python float %: result takes the divisor's sign
    const int one = 65536;
    if (angle == 0.0) return 0;
    if (angle == 180.0) {
        out[0] = Op{{MISEG_AUG_VFLIP, 0, 0, 0, 0, 0, 0, w, h}};
        out[1] = Op{{MISEG_AUG_HFLIP, 0, 0, 0, 0, 0, 0, w, h}};
        return 2;
    }
    if (angle == 90.0 && w == h) {
        out[0] = Op{{MISEG_AUG_AFFINE, 0, -one, (w - 1) * one + one / 2, one, 0, one / 2, w, h}};
        return 1;
    }
    if (angle == 270.0 && w == h) {
        out[0] = Op{{MISEG_AUG_AFFINE, 0, one, one / 2, -one, 0, (h - 1) * one + one / 2, w, h}};
        return 1;
    }
    const double rad = -(angle * (M_PI / 180.0));          // math.radians
    double m[6] = {py_round15(cos(rad)), py_round15(sin(rad)), 0.0, py_round15(-sin(rad)), py_round15(cos(rad)), 0.0};
    const double cx = w / 2.0, cy = h / 2.0;
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2];
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5];
    m[2] += cx;
    m[5] += cy;
    out[0] = Op{{MISEG_AUG_AFFINE, fix16(m[0]), fix16(m[1]), fix16(m[2] + m[0] * 0.5 + m[1] * 0.5), fix16(m[3]), fix16(m[4]),
                 fix16(m[5] + m[3] * 0.5 + m[4] * 0.5), w, h}};
    return 1;
}

// One SequentialWrapper.__call__: comm transform under seed `comm`, image transform under seed `img`.
int plan_view(const miseg_aug_recipe* rc, PyRandom& rng, PyRandom& r2, int slice, int w, int h, int32_t* row, int* ow, int* oh) {
    for (int i = 0; i < MISEG_AUG_JOB_INTS; ++i) row[i] = 0;
    Op ops[MISEG_AUG_MAX_GEO + 2];
    int n = 0;
    for (int g = 0; g < rc->n_geo; ++g) {
        if (n > MISEG_AUG_MAX_GEO) break;
        const double arg = rc->geo_arg[g];
        switch (rc->geo_kind[g]) {
            case MISEG_RECIPE_ROTATE: n += rotation_ops(rng.uniform(-arg, arg), w, h, ops + n); break;
            case MISEG_RECIPE_VFLIP:
                if (rng.random() < arg) ops[n++] = Op{{MISEG_AUG_VFLIP, 0, 0, 0, 0, 0, 0, w, h}};
                break;
            case MISEG_RECIPE_HFLIP:
                if (rng.random() < arg) ops[n++] = Op{{MISEG_AUG_HFLIP, 0, 0, 0, 0, 0, 0, w, h}};
                break;
            case MISEG_RECIPE_RANDOM_CROP: {
                const int t = (int)arg;
                int i = 0, j = 0;
                if (!(w == t && h == t)) {
                    if (h < t || w < t) return miseg::fail(MISEG_E_INVALID, "plan_augment: RandomCrop(%d) on a %dx%d slice", t, w, h);
                    i = rng.randint(0, h - t);
                    j = rng.randint(0, w - t);
                }
                ops[n++] = Op{{MISEG_AUG_CROP, i, j, 0, 0, 0, 0, w, h}};
                w = h = t;
                break;
            }
            case MISEG_RECIPE_CENTER_CROP: {
                const int t = (int)arg;
                const int i = (int)nearbyint((h - t) / 2.0), j = (int)nearbyint((w - t) / 2.0);  // Python round(): half to even
                ops[n++] = Op{{MISEG_AUG_CROP, i, j, 0, 0, 0, 0, w, h}};
                w = h = t;
                break;
            }
            default: return miseg::fail(MISEG_E_INVALID, "plan_augment: unknown geometric kind %d", rc->geo_kind[g]);
        }
    }
    if (n > MISEG_AUG_MAX_GEO) return miseg::fail(MISEG_E_INVALID, "plan_augment: %d geometric ops, the job table holds %d", n, MISEG_AUG_MAX_GEO);
    row[0] = slice;
    row[3] = n;
    for (int g = 0; g < n; ++g)
        for (int k = 0; k < 9; ++k) row[12 + 9 * g + k] = ops[g].v[k];
    if (rc->has_jitter) {  // torchvision 0.7 ColorJitter.get_params: uniform per factor, then random.shuffle
        int code[3] = {MISEG_AUG_BRIGHTNESS, MISEG_AUG_CONTRAST, MISEG_AUG_SATURATION};
        double f[3];
        for (int c = 0; c < 3; ++c) f[c] = r2.uniform(rc->jitter[2 * c], rc->jitter[2 * c + 1]);
        for (int i = 2; i >= 1; --i) {
            const int j = (int)r2.randbelow((uint32_t)i + 1);
            const int tc = code[i]; code[i] = code[j]; code[j] = tc;
            const double tf = f[i]; f[i] = f[j]; f[j] = tf;
        }
        row[4] = 3;
        for (int c = 0; c < 3; ++c) {
            row[5 + c] = code[c];
            const float a = (float)f[c];
            memcpy(&row[8 + c], &a, 4);
        }
    }
    *ow = w;
    *oh = h;
    return MISEG_OK;
}

}  // namespace

extern "C" int miseg_plan_augment(const miseg_aug_recipe* recipe, int64_t n_items, const int64_t* item_seeds,
                                  const int32_t* slice_ids, const int32_t* widths, const int32_t* heights, int32_t* jobs_out,
                                  int32_t* out_wh) {
    MISEG_REQUIRE(recipe && item_seeds && slice_ids && widths && heights && jobs_out && out_wh, "plan_augment: null pointer");
    MISEG_REQUIRE(n_items >= 0 && recipe->n_geo >= 0 && recipe->n_geo <= MISEG_AUG_MAX_GEO, "plan_augment: bad sizes");
    const int views = recipe->twice ? 2 : 1;
    int ow = -1, oh = -1;
    std::vector<PyRandom> item_rng((size_t)n_items), view_rng((size_t)n_items * views * 2);
    std::vector<uint64_t> seeds((size_t)n_items), vseeds((size_t)n_items * views * 2);
    for (int64_t it = 0; it < n_items; ++it) {
        MISEG_REQUIRE(item_seeds[it] >= 0, "plan_augment: negative item seed");
        seeds[it] = (uint64_t)item_seeds[it];
    }
    seed_batch(seeds.data(), (int)n_items, item_rng.data());
    for (int64_t it = 0; it < n_items; ++it) {  // SequentialWrapperTwice: six randint(0, 1e5) under FixRandomSeed(global_seed)
        PyRandom& rng = item_rng[it];
        uint64_t comm[2], img[2];
        if (recipe->twice) {
            comm[0] = rng.randint(0, 100000), comm[1] = rng.randint(0, 100000);
            img[0] = rng.randint(0, 100000), img[1] = rng.randint(0, 100000);
            if (!recipe->total_freedom) comm[1] = comm[0];
        } else {
            comm[0] = rng.randint(0, 100000), img[0] = rng.randint(0, 100000);
        }
        for (int v = 0; v < views; ++v) {
            vseeds[((size_t)it * views + v) * 2] = comm[v];
            vseeds[((size_t)it * views + v) * 2 + 1] = img[v];
        }
    }
    seed_batch(vseeds.data(), (int)vseeds.size(), view_rng.data());
    for (int64_t it = 0; it < n_items; ++it) {
        for (int v = 0; v < views; ++v) {
            int w = 0, h = 0;
            const size_t s = ((size_t)it * views + v) * 2;
            int rc = plan_view(recipe, view_rng[s], view_rng[s + 1], slice_ids[it], widths[it], heights[it],
                               jobs_out + ((int64_t)v * n_items + it) * MISEG_AUG_JOB_INTS, &w, &h);
            if (rc != MISEG_OK) return rc;
            if (ow < 0) ow = w, oh = h;
            MISEG_REQUIRE(w == ow && h == oh, "plan_augment: slices of different output sizes in one batch (%dx%d vs %dx%d)", w, h, ow, oh);
        }
    }
    out_wh[0] = ow, out_wh[1] = oh;
    return MISEG_OK;
}
