#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <random>
extern "C" int mvf_kdtree_order(const float *, int64_t, int32_t, int64_t, int64_t, int64_t *, int32_t);
int main() {
    std::mt19937 rng(5);
    std::normal_distribution<float> nd;
    for (int rep = 0; rep < 40; ++rep) {
        int64_t n = 1 + (rng() % 200000);
        if (rep < 8) n = 1 + rep * 37;
        std::vector<float> p(3 * n);
        int kind = rep % 4;
        for (auto &v : p) v = kind == 0 ? nd(rng) : kind == 1 ? (float)(rng() % 4) : kind == 2 ? 1.f : nd(rng) * 0.05f;
        std::vector<int64_t> a(n), b(n);
        int dims = 2 + (rep & 1);
        if (mvf_kdtree_order(p.data(), n, dims, 3, 50, a.data(), 1)) return 1;
        if (mvf_kdtree_order(p.data(), n, dims, 3, 50, b.data(), 2 + rep % 7)) return 2;
        for (int64_t i = 0; i < n; ++i) if (a[i] != b[i]) { printf("mismatch rep %d at %lld\n", rep, (long long)i); return 3; }
    }
    printf("ok\n");
    return 0;
}
